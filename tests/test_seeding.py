"""seeding (mrcal_amd/calibration.py; reference:
mrcal/calibration.py:508-1610, traverse-sensor-links.c) - the step before the
optimizer - and the whole chain seed -> staged optimize() the way
mrcal-calibrate-cameras:404-826 runs it.

CPU (the geometry, on exact synthetic vectors):
  - planar pose from observation vectors: exact data, noisy data, masked
    corners, boards partly BEHIND the image plane (where the reference's
    pinhole + solvePnP route needs its focal-length retries), too few corners
  - Procrustes alignment, the traversal order of the sensor graph (fewest
    hops, then most shared frames; unreachable sensors are not visited)
  - camera poses chained through a camera that never sees a frame together
    with camera 0; frame poses merged over the cameras that see the frame
GPU:
  - seed_stereographic + the staged solves on a synthetic 3-camera rig: the
    truth comes back
  - the same chain on the REAL calibration data of tests/golden: from nothing
    but the detected corners and a focal-length guess to the stored model's
    intrinsics and fit
"""
import os
import numpy as np
import pytest

from conftest import GOLDEN_DIR


def board_in_view(rng, N, wide=False):
    from mrcal_amd import poseutils as pu
    rt = np.concatenate((rng.normal(size=(N, 3))*(0.6 if wide else 0.3),
                         rng.normal(size=(N, 3))*np.array((0.5, 0.5, 0.3)) + np.array((-0.45, -0.45, 0.9 if wide else 2.5))), -1)
    return pu.Rt_from_rt(rt)


def test_planar_pose_from_vectors():
    from mrcal_amd import calibration as cal, poseutils as pu
    rng = np.random.default_rng(0)
    obj = cal.ref_calibration_object(10, 8, 0.1).reshape(-1, 3)
    assert obj.shape == (80, 3) and np.allclose(obj[1], (0.1, 0, 0)) and np.allclose(obj[10], (0, 0.1, 0))
    for wide in (False, True):
        Rt = board_in_view(rng, 300, wide)
        p = pu.transform_point_Rt(Rt[:, None], obj[None])
        if wide: assert (p[..., 2] < 0).any()              # corners behind the image plane of any pinhole
        v = p*rng.uniform(0.3, 3., size=p.shape[:2])[..., None]                # any length
        # exact data: the pose to rounding
        est, rms = cal._planar_pose_from_vectors(v, obj[:, :2], np.ones(p.shape[:2], bool))
        assert np.abs(pu.transform_point_Rt(est[:, None], obj[None]) - p).max() < 1e-7
        assert rms.max() < 1e-8
        assert np.abs(np.linalg.det(est[:, :3, :]) - 1).max() < 1e-12
        # 1 mrad of noise, a fifth of the corners missing: centimetre-level at a metre
        vn = v/np.linalg.norm(v, axis=-1, keepdims=True) + rng.normal(size=v.shape)*1e-3
        mask = rng.uniform(size=v.shape[:2]) > 0.2
        vn[~mask] = np.nan                                   # masked corners may hold anything
        est, rms = cal._planar_pose_from_vectors(np.nan_to_num(vn), obj[:, :2], mask)
        err = np.abs(pu.transform_point_Rt(est[:, None], obj[None]) - p).max((1, 2))
        assert np.median(err) < 1e-2 and err.max() < 0.1         # (depth along the line of sight is the weak direction)
        assert rms.max() < 3e-3
    with pytest.raises(Exception, match="Insufficient"):
        m = np.zeros((1, 80), bool); m[0, :3] = True
        cal._planar_pose_from_vectors(v[:1], obj[:, :2], m)


def test_procrustes_and_traversal():
    from mrcal_amd import calibration as cal, poseutils as pu
    rng = np.random.default_rng(1)
    Rt = pu.Rt_from_rt(np.array((0.3, -0.2, 0.5, 1., 2., -3.)))
    B = rng.normal(size=(40, 3))
    A = pu.transform_point_Rt(Rt, B)
    assert np.abs(cal.align_procrustes_points_Rt01(A, B) - Rt).max() < 1e-12
    assert np.abs(cal.align_procrustes_points_Rt01(A + rng.normal(size=A.shape)*1e-3, B) - Rt).max() < 2e-3
    # a reflection is not a rotation
    Rt2 = cal.align_procrustes_points_Rt01(A*np.array((1, 1, -1.)), B)
    assert abs(np.linalg.det(Rt2[:3]) - 1) < 1e-12
    with pytest.raises(Exception): cal.align_procrustes_points_Rt01(np.outer(np.arange(5.), (1, 2, 3)), np.outer(np.arange(5.), (1, 0, 0)))

    # 0-1 (5 frames), 0-2 (2), 1-2 (50), 2-3 (9), 1-3 (3); 4 is alone
    C = np.zeros((5, 5), int)
    for a, b, n in ((0, 1, 5), (0, 2, 2), (1, 2, 50), (2, 3, 9), (1, 3, 3)): C[a, b] = C[b, a] = n
    links = []
    cal.traverse_sensor_links(connectivity_matrix=C, callback_sensor_link=lambda i, f: links.append((i, f)))
    # one hop beats two however many frames the detour shares; among equal hop counts the most shared frames win;
    # nearer sensors are reported first, the better-connected of two equally near ones first
    assert links == [(1, 0), (2, 0), (3, 2)]
    with pytest.raises(Exception): cal.traverse_sensor_links(connectivity_matrix=np.triu(C), callback_sensor_link=lambda i, f: None)


def test_camera_and_frame_poses_from_exact_monocular_poses():
    from mrcal_amd import calibration as cal, poseutils as pu
    rng = np.random.default_rng(2)
    W, H, sp = 9, 7, 0.05
    Ncam, Nframes = 4, 30
    rt_cam_ref = np.concatenate((np.zeros((1, 6)), np.concatenate((rng.normal(size=(Ncam-1, 3))*0.1,
                                                                   rng.normal(size=(Ncam-1, 3))*0.3), -1)))
    rt_ref_frame = pu.rt_from_Rt(board_in_view(rng, Nframes))
    # camera 3 never shares a frame with camera 0: it hangs off camera 2; some frames are seen by one camera only
    idx = []
    for f in range(Nframes):
        cams = ((0, 1), (0, 1, 2), (2, 3), (1, 2, 3), (3,), (0,))[f % 6]
        idx += [(f, c) for c in cams]
    idx = np.array(idx, dtype=np.int32)
    Rt_cf = pu.compose_Rt(pu.Rt_from_rt(rt_cam_ref[idx[:, 1]]), pu.Rt_from_rt(rt_ref_frame[idx[:, 0]]))
    Rt_0c = cal._estimate_camera_poses(Rt_cf, idx, W, H, sp)
    assert Rt_0c.shape == (3, 4, 3)
    assert np.abs(pu.invert_Rt(Rt_0c) - pu.Rt_from_rt(rt_cam_ref[1:])).max() < 1e-10
    rt_rf = cal.estimate_joint_frame_poses(Rt_cf, pu.invert_Rt(Rt_0c), idx, W, H, sp)
    assert rt_rf.shape == (Nframes, 6)
    assert np.abs(pu.Rt_from_rt(rt_rf) - pu.Rt_from_rt(rt_ref_frame)).max() < 1e-10
    # noisy monocular poses: the merged frame pose is no worse than the observations it came from
    noisy = pu.compose_Rt(pu.Rt_from_rt(rng.normal(size=(len(idx), 6))*1e-3), Rt_cf)
    rt_rf2 = cal.estimate_joint_frame_poses(noisy, pu.Rt_from_rt(rt_cam_ref[1:]), idx, W, H, sp)
    assert np.abs(pu.Rt_from_rt(rt_rf2) - pu.Rt_from_rt(rt_ref_frame)).max() < 1e-2
    # no path to camera 0
    lonely = idx[~((idx[:, 1] == 3) & np.isin(idx[:, 0], idx[idx[:, 1] == 2, 0]))]
    lonely = lonely[~((lonely[:, 1] == 3) & np.isin(lonely[:, 0], lonely[lonely[:, 1] == 1, 0]))]
    sel = np.array([np.nonzero(np.all(idx == row, axis=1))[0][0] for row in lonely])
    with pytest.raises(Exception, match="overlap"):
        cal._estimate_camera_poses(Rt_cf[sel], lonely, W, H, sp)
    # frames out of order
    with pytest.raises(Exception): cal._estimate_camera_poses(Rt_cf[::-1], idx[::-1], W, H, sp)


# ----------------------------------------------------------------------------- seed -> optimize
def calibrate_like_the_tool(amd, imagersizes, focal, indices_frame_camera, observations, spacing, lensmodel):
    """mrcal-calibrate-cameras:404-826 in brief: seed; geometry only; + the
    stereographic core; the full model without the board warp; everything"""
    from mrcal_amd.calibration import seed_stereographic
    core, rt_cam_ref, rt_ref_frame = seed_stereographic(imagersizes, focal, indices_frame_camera, observations, spacing)
    idx = np.ascontiguousarray(np.concatenate((indices_frame_camera, indices_frame_camera[:, 1:] - 1), -1).astype(np.int32))
    seed = dict(intrinsics=core.copy(), rt_cam_ref=rt_cam_ref.copy(), rt_ref_frame=rt_ref_frame.copy())
    oi = dict(intrinsics=core, rt_cam_ref=rt_cam_ref, rt_ref_frame=rt_ref_frame,
              observations_board=observations.copy(), indices_frame_camintrinsics_camextrinsics=idx,
              lensmodel="LENSMODEL_STEREOGRAPHIC", imagersizes=np.asarray(imagersizes, dtype=np.int32),
              do_optimize_intrinsics_core=False, do_optimize_intrinsics_distortions=False, do_optimize_calobject_warp=False,
              calibration_object_spacing=spacing, do_apply_outlier_rejection=False, do_apply_regularization=False)
    rms = [amd.optimize(**oi)["rms_reproj_error__pixels"]]
    oi["do_optimize_intrinsics_core"] = True
    rms.append(amd.optimize(**oi)["rms_reproj_error__pixels"])
    N = amd.lensmodel_num_params(lensmodel)
    rng = np.random.RandomState(0)
    extra = (rng.random_sample((len(imagersizes), N - 4)) - 0.5)*2e-6
    if "OPENCV8" in lensmodel or "OPENCV12" in lensmodel: extra[:, 5:8] *= 1e-3
    oi.update(intrinsics=np.ascontiguousarray(np.concatenate((oi["intrinsics"], extra), -1)), lensmodel=lensmodel,
              do_optimize_intrinsics_distortions=True, do_apply_outlier_rejection=True, do_apply_regularization=True)
    rms.append(amd.optimize(**oi)["rms_reproj_error__pixels"])
    oi.update(calobject_warp=np.zeros(2), do_optimize_calobject_warp=True)
    seed["last_stage_inputs"] = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in oi.items()}
    stats = amd.optimize(**oi)
    rms.append(stats["rms_reproj_error__pixels"])
    return oi, stats, rms, seed


@pytest.mark.gpu
def test_seed_and_calibrate_synthetic_rig(amd):
    from mrcal_amd.synthetic import make_calibration_problem
    from mrcal_amd import poseutils as pu
    oi0, truth = make_calibration_problem(amd._api, Ncameras=3, Nframes=40, lensmodel="LENSMODEL_OPENCV4", seed=7,
                                          pixel_noise=0.3, make_outliers=False)
    obs = oi0["observations_board"]
    idx = oi0["indices_frame_camintrinsics_camextrinsics"][:, :2].copy()
    f_true = truth["intrinsics"][0, 0]
    oi, stats, rms, seed = calibrate_like_the_tool(amd, oi0["imagersizes"], f_true*1.15, idx, obs, oi0["calibration_object_spacing"],
                                                   "LENSMODEL_OPENCV4")
    # the seed alone: the right neighbourhood, no more (15 % focal-length error in, no distortions, the centre pixel
    # assumed in the middle of the imager: the boards land at the wrong depth and the cameras with them)
    t_true = truth["rt_cam_ref"][:, 3:]
    assert np.abs(seed["rt_cam_ref"][:, 3:] - t_true).max() < 1.0
    assert np.abs(seed["rt_cam_ref"][:, :3] - truth["rt_cam_ref"][:, :3]).max() < 0.2
    # the chain: every stage fits better than the one before, and the end is the truth
    assert rms[1] <= rms[0] and rms[2] < rms[1] and rms[3] <= rms[2] + 1e-9
    assert stats["rms_reproj_error__pixels"] < 0.35                      # 0.3 px of noise went in
    assert np.abs(oi["intrinsics"][:, :4] - truth["intrinsics"][:, :4]).max() < 2.0
    assert np.abs(oi["rt_cam_ref"][:, 3:] - t_true).max() < 5e-3
    assert np.abs(oi["rt_cam_ref"][:, :3] - truth["rt_cam_ref"][:, :3]).max() < 2e-3
    assert np.abs(oi["calobject_warp"] - truth["calobject_warp"]).max() < 1e-3


@pytest.mark.gpu
def test_seed_and_calibrate_real_data(amd, ref_api):
    """nothing but the detected corners of the real calibration and a rough
    focal length: the chain must arrive at the model that is stored with them"""
    from mrcal_amd.cameramodel import cameramodel
    m = cameramodel(os.path.join(GOLDEN_DIR, "real_opencv8-0.cameramodel"))
    stored = m.optimization_inputs()
    obs = stored["observations_board"].copy()
    # (the stored weights carry the original solve's outlier marks, which cannot be told from corners the detector
    #  never saw: both are negative. They stay; outlier rejection runs on top of them as it would in the tool)
    idx = stored["indices_frame_camintrinsics_camextrinsics"][:, :2].copy()
    oi, stats, rms, seed = calibrate_like_the_tool(amd, stored["imagersizes"], 1900., idx, obs, stored["calibration_object_spacing"],
                                                   "LENSMODEL_OPENCV8")
    ref = amd.optimize(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in stored.items()})
    # (0.466 px stored; the chain ends 0.47-0.49 px. Five solves with outlier rejection in a row are a cascade of
    #  k-sigma thresholds: a change in the last bits of one sum - the order a reduction runs in - moves a corner or
    #  two across a threshold and the rms by 1e-2 px. What the chain is held to is the model it arrives at)
    assert abs(stats["rms_reproj_error__pixels"] - ref["rms_reproj_error__pixels"]) < 0.05
    # ... and the tight bound sits on a deterministic quantity: the chain's LAST solve, from the very
    # state and outlier marks the chain handed it, by the reference's own mrcal_optimize(): the same corners thrown
    # out, the rms to 1e-6 relative - the cascade above is the chain's, not the solver's
    last = seed["last_stage_inputs"]
    o_r = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in last.items()}
    s_r = ref_api.optimize(**o_r)
    # (round 5: "the same corners" up to the ones AT the threshold. The library's arithmetic changed in its last bits
    #  (-ffp-contract=on, csrc/build.sh) and with it two of 18 600 corners fell on the other side of the k-sigma line in
    #  this solve's passes: 671 against the reference's 669. Where the marks agree the rms agrees to 1e-6; where a
    #  handful differ it is the rms of a slightly different problem)
    mask_r, mask_a = o_r["observations_board"][...,2] < 0, oi["observations_board"][...,2] < 0
    ndiff = int((mask_r != mask_a).sum())
    assert ndiff <= 10, ndiff          # (6 observed: four corners traded places, two more on the product's side)
    assert abs(s_r["Noutliers_board"] - stats["Noutliers_board"]) <= ndiff
    # what is NOT a threshold effect stays strict. (1) every corner the two solves mark differently sits AT
    # the k-sigma line (markOutliers: k0 = 4, /root/reference/mrcal.c:3978-4402): seen from the solve that keeps it, its
    # residual is 3.5 - 4.6 sigma of that solve's inliers - a corner of 2 or of 10 sigma marked by one and not by the
    # other would be a defect of the marking, not rounding. (2) on the corners BOTH solves keep, the two solutions fit
    # the same: rms over the common inliers to 1e-3 (observed 2.8e-4, the product's the LOWER one: a handful of 4-sigma
    # corners in or out of 18 000 move the optimum, and the restated libdogleg stops a little earlier in OPENCV8's flat
    # valley; 1e-2 on the whole-problem rms, which this replaces, would have passed a real regression)
    Ncorn = mask_a.size
    xa = stats["x"][:2*Ncorn].reshape(-1, 2); xr = s_r["x"][:2*Ncorn].reshape(-1, 2)
    w  = np.abs(last["observations_board"][..., 2].ravel())             # (|weight|: the residuals are weighted)
    both = ~mask_a.ravel() & ~mask_r.ravel()
    rms_a = np.sqrt((xa[both]**2).sum()/(2*both.sum())); rms_r = np.sqrt((xr[both]**2).sum()/(2*both.sum()))
    print(f"{ndiff} corners marked differently; rms over the {int(both.sum())} common inliers: {rms_a!r} (product) {rms_r!r} (reference)")
    assert abs(rms_a - rms_r) < 1e-3*rms_r, (rms_a, rms_r)
    for i in np.nonzero((mask_a != mask_r).ravel())[0]:
        x_keep, sigma = (xr[i], rms_r) if mask_a.ravel()[i] else (xa[i], rms_a)      # the solve that keeps corner i as an inlier
        nsig = np.abs(x_keep).max()/sigma
        print(f"   corner {i}: kept by the {'reference' if mask_a.ravel()[i] else 'product'} at {nsig:.3f} sigma (weight {w[i]:.3f})")
        assert 3.5 < nsig < 4.6, (i, nsig)
    if ndiff == 0:
        assert abs(s_r["rms_reproj_error__pixels"] - stats["rms_reproj_error__pixels"]) < 1e-6*s_r["rms_reproj_error__pixels"]
    assert np.abs(oi["intrinsics"][0, :4] - stored["intrinsics"][0, :4]).max() < 3.0       # pixels, on a 6016x4016 imager
    assert np.abs(oi["calobject_warp"] - stored["calobject_warp"]).max() < 5e-4
