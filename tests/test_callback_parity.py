"""PARITY of the HIP optimizer_callback() path against the reference.

  - the six golden cases of the reference's test/test-optimizer-callback.py
    (committed fixtures, tests/golden/): x and J
  - seeded synthetic problems over every supported lens model and
    do_optimize_* combination, against the reference's own C code
    (oracle/_ref) on identical inputs

Bars: CSR structure (rowptr, colidx), Nstate/Nmeas/Nnz and b_packed BIT-EXACT;
x and J values within 1e-6 relative (the reference's own relative-error
definition, test/testutils.py:105), which is what north_star asks for. In
practice we are within ~1e-12."""
import numpy as np
import pytest

from conftest import golden_case_inputs, relative_error
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6


def compare_callbacks(res_amd, res_ref, what=""):
    b_a, x_a, J_a, _ = res_amd
    b_r, x_r, J_r, _ = res_ref
    assert np.array_equal(b_a, b_r), f"{what}: b_packed differs"
    assert x_a.shape == x_r.shape
    err = relative_error(x_a, x_r)
    assert err.max() < REL_TOL, f"{what}: x rel err {err.max()} at {err.argmax()}"
    if J_r is None:
        assert J_a is None
        return
    assert J_a.shape == J_r.shape
    assert np.array_equal(J_a.indptr,  J_r.indptr),  f"{what}: CSR rowptr differs"
    assert np.array_equal(J_a.indices, J_r.indices), f"{what}: CSR colidx differs"
    err = relative_error(J_a.data, J_r.data)
    i = err.argmax()
    assert err.max() < REL_TOL, \
        f"{what}: J rel err {err.max()} at nnz {i}: ours {J_a.data[i]} ref {J_r.data[i]}"


@pytest.mark.parametrize("icase", range(6))
def test_golden_vectors(amd, golden, icase):
    kw = golden_case_inputs(golden, icase)
    b, x, J, _ = amd.optimizer_callback(no_factorization=True, **kw)
    Jd = J.toarray()
    amd.pack_state(Jd, **kw)

    n = 810 if icase in (0,1,3) else x.size
    # the reference's shipped vectors (observation rows)
    assert relative_error(x[:n], golden[f"x_ref_{icase}"][:n]).max() < REL_TOL
    assert relative_error(Jd[:n], golden[f"J_ref_{icase}"][:n]).max() < REL_TOL
    # the reference library's current output (all rows)
    assert np.array_equal(b, golden[f"b_lib_{icase}"])
    assert relative_error(x,  golden[f"x_lib_{icase}"]).max() < REL_TOL
    assert relative_error(Jd, golden[f"J_lib_{icase}"]).max() < REL_TOL
    # sparsity pattern: the dense golden has a nonzero wherever we store one,
    # except explicitly-stored zeros (outliers, fy column of an x row, ...)
    assert np.all( (golden[f"J_lib_{icase}"] != 0) <= (np.abs(J).toarray() >= 0) )


def test_golden_no_jacobian(amd, golden):
    kw = golden_case_inputs(golden, 4)
    b, x, J, f = amd.optimizer_callback(no_jacobian=True, no_factorization=True, **kw)
    assert J is None and f is None
    assert relative_error(x, golden["x_lib_4"]).max() < REL_TOL


LENSMODELS = ("LENSMODEL_PINHOLE", "LENSMODEL_STEREOGRAPHIC", "LENSMODEL_LONLAT", "LENSMODEL_LATLON",
              "LENSMODEL_OPENCV4", "LENSMODEL_OPENCV5", "LENSMODEL_OPENCV8", "LENSMODEL_OPENCV12",
              "LENSMODEL_CAHVOR", "LENSMODEL_CAHVORE_linearity=0.00", "LENSMODEL_CAHVORE_linearity=0.37",
              "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120",
              "LENSMODEL_SPLINED_STEREOGRAPHIC_order=2_Nx=9_Ny=7_fov_x_deg=110")


@pytest.mark.parametrize("lensmodel", LENSMODELS)
def test_synthetic_all_variables(amd, ref_api, lensmodel):
    oi, _ = make_calibration_problem(amd._api, Ncameras=3, Nframes=7, lensmodel=lensmodel,
                                     object_width_n=9, object_height_n=8, seed=3)
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                      ref_api.optimizer_callback(no_factorization=True, **oi), lensmodel)


def _with_points(oi, rng, Npoints=6, Npoints_fixed=2):
    """adds discrete-point observations (incl. an outlier and fixed points)"""
    oi = copy_inputs(oi)
    Ncam = oi["intrinsics"].shape[0]
    oi["points"] = np.ascontiguousarray(rng.uniform(-1,1,size=(Npoints,3))*np.array((1.,1.,0.5)) +
                                        np.array((0.3,0,5.)))
    idx = []
    for ip in range(Npoints):
        for ic in rng.choice(Ncam, size=min(2,Ncam), replace=False):
            idx.append((ip, ic, ic-1))
    idx = np.array(idx, dtype=np.int32)
    obs = np.zeros((len(idx),3))
    obs[:,:2] = rng.uniform(500,1800,size=(len(idx),2))
    obs[:,2]  = rng.uniform(0.5,1.5,size=len(idx))
    obs[1,2]  = -1.    # outlier
    obs[2,2]  = 0.     # weight exactly 0: an outlier for points (mrcal.c:4918)
    oi["observations_point"] = obs
    oi["indices_point_camintrinsics_camextrinsics"] = idx
    oi["Npoints_fixed"] = Npoints_fixed
    return oi


def test_synthetic_selections_and_points(amd, ref_api):
    rng = np.random.RandomState(7)
    oi0, _ = make_calibration_problem(amd._api, Ncameras=2, Nframes=5, lensmodel="LENSMODEL_OPENCV8",
                                      object_width_n=10, object_height_n=10, seed=5)
    oi0 = _with_points(oi0, rng)
    # some input outliers on the boards, and a weight of exactly 0
    oi0["observations_board"][1,2:5,3:6,2] = -1.
    oi0["observations_board"][3,0,0,2]     = 0.
    import itertools
    flags = ("do_optimize_intrinsics_core", "do_optimize_intrinsics_distortions",
             "do_optimize_extrinsics", "do_optimize_frames", "do_optimize_calobject_warp",
             "do_apply_regularization", "do_apply_regularization_unity_cam01")
    for bits in itertools.product((False,True), repeat=len(flags)):
        if not any(bits[:5]):
            continue
        if rng.rand() < 0.6:
            continue
        oi = copy_inputs(oi0)
        oi.update(dict(zip(flags, bits)))
        compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                          ref_api.optimizer_callback(no_factorization=True, **oi), str(bits))


@pytest.mark.parametrize("Ncam,Nf,W,H,lensmodel", ((1, 1, 2, 2, "LENSMODEL_PINHOLE"),
                                                   (2, 1, 3, 2, "LENSMODEL_OPENCV4"),
                                                   (1, 3, 13, 11, "LENSMODEL_OPENCV8"),        # 143 corners: three passes
                                                   (20, 3, 5, 4, "LENSMODEL_OPENCV4")))        # 276 camera-block variables
def test_edge_shapes(amd, ref_api, Ncam, Nf, W, H, lensmodel):
    """the smallest boards, one frame, an observation that is all outliers, a weight of
    exactly 0, many cameras"""
    oi, _ = make_calibration_problem(amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel=lensmodel,
                                     object_width_n=W, object_height_n=H, seed=11, make_outliers=False)
    oi["observations_board"][0,:,:,2]  = -1.       # the first observation: nothing but outliers
    oi["observations_board"][-1,0,0,2] = 0.        # a weight of exactly 0
    oi["do_apply_outlier_rejection"] = False
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **copy_inputs(oi)),
                      ref_api.optimizer_callback(no_factorization=True, **copy_inputs(oi)), f"{Ncam}x{Nf} {W}x{H}")
    if Ncam >= 20:
        # ... and the solve, through the panel-by-panel Cholesky of a camera block that does not fit the LDS
        oa, orr = copy_inputs(oi), copy_inputs(oi)
        sa, sr = amd.optimize(**oa), ref_api.optimize(**orr)
        assert abs(sa["rms_reproj_error__pixels"] - sr["rms_reproj_error__pixels"]) < 1e-6*sr["rms_reproj_error__pixels"]


def points_only_problem(api, seed=3, Ncam=3, Np=30, Nfixed=4, lens="LENSMODEL_OPENCV4"):
    """discrete points seen by every camera, NO chessboards: extrinsics and the free
    points are optimized against a few fixed points (the gauge)"""
    from mrcal_amd.synthetic import intrinsics_for, IMAGERSIZE, R_from_r
    rng  = np.random.RandomState(seed)
    intr = intrinsics_for(lens, Ncam)
    rt = np.zeros((Ncam-1,6))
    rt[:,:3] = rng.uniform(-0.05, 0.05, (Ncam-1,3))
    rt[:,3]  = -0.3*np.arange(1, Ncam)
    rt[:,4:] = rng.uniform(-0.05, 0.05, (Ncam-1,2))
    pts = rng.uniform(-1, 1, (Np,3))*np.array((1.5, 1.0, 1.0)) + np.array((0.3, 0, 5.))
    idx = np.array([(ip, ic, ic-1) for ip in range(Np) for ic in range(Ncam)], dtype=np.int32)
    q = np.zeros((len(idx),3))
    for n, (ip, ic, ie) in enumerate(idx):
        p = pts[ip] if ie < 0 else R_from_r(rt[ie,:3]) @ pts[ip] + rt[ie,3:]
        q[n,:2] = api.project(p[None], lens, intr[ic])[0]
    q[:,:2] += rng.normal(0, 0.3, (len(idx),2))
    q[:,2]   = rng.uniform(0.5, 1.5, len(idx))
    return dict(intrinsics=intr.copy(), rt_cam_ref=rt + rng.normal(0, 0.01, rt.shape), rt_ref_frame=None,
                points=pts + np.r_[rng.normal(0, 0.05, (Np-Nfixed,3)), np.zeros((Nfixed,3))],
                observations_board=None, indices_frame_camintrinsics_camextrinsics=None,
                observations_point=q, indices_point_camintrinsics_camextrinsics=idx,
                lensmodel=lens, imagersizes=np.array((IMAGERSIZE,)*Ncam, dtype=np.int32),
                do_optimize_intrinsics_core=False, do_optimize_intrinsics_distortions=False,
                do_optimize_extrinsics=True, do_optimize_frames=True,
                do_optimize_calobject_warp=False, calobject_warp=None, calibration_object_spacing=0.,
                do_apply_regularization=False, do_apply_outlier_rejection=False, Npoints_fixed=Nfixed, verbose=False)


def test_points_only_no_boards(amd, ref_api):
    """no board observations at all: the point kernels, 3x3 blocks only in the
    Schur elimination, no frames"""
    oi = points_only_problem(ref_api)
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **copy_inputs(oi)),
                      ref_api.optimizer_callback(no_factorization=True, **copy_inputs(oi)), "points only")
    oa, orr = copy_inputs(oi), copy_inputs(oi)
    sa, sr = amd.optimize(**oa), ref_api.optimize(**orr)
    assert abs(sa["rms_reproj_error__pixels"] - sr["rms_reproj_error__pixels"]) < 1e-6*sr["rms_reproj_error__pixels"]
    assert np.abs(oa["points"] - orr["points"]).max() < 1e-5
    assert np.abs(oa["rt_cam_ref"] - orr["rt_cam_ref"]).max() < 1e-5
    assert np.array_equal(oa["points"][-4:], oi["points"][-4:])          # the fixed points stay put


@pytest.mark.parametrize("lensmodel", ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120",
                                       "LENSMODEL_CAHVORE_linearity=0.37"))
def test_points_and_selections_other_models(amd, ref_api, lensmodel):
    """discrete points (incl. outliers: the splined model names its first
    (order+1)^2 knots for those, mrcal.c:4960-4972), board outliers and a few
    do_optimize_* combinations for the non-OPENCV kernels"""
    rng = np.random.RandomState(11)
    oi0, _ = make_calibration_problem(amd._api, Ncameras=2, Nframes=4, lensmodel=lensmodel,
                                      object_width_n=7, object_height_n=6, seed=6)
    oi0 = _with_points(oi0, rng)
    oi0["observations_board"][1,2:4,3:5,2] = -1.
    for core, dist, ext, frames, warp, reg in ((1,1,1,1,1,1), (0,1,1,1,0,1), (1,0,0,1,1,0), (0,1,0,0,0,1), (1,1,1,0,1,1)):
        oi = copy_inputs(oi0)
        oi.update(do_optimize_intrinsics_core=bool(core), do_optimize_intrinsics_distortions=bool(dist),
                  do_optimize_extrinsics=bool(ext), do_optimize_frames=bool(frames),
                  do_optimize_calobject_warp=bool(warp), do_apply_regularization=bool(reg))
        compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                          ref_api.optimizer_callback(no_factorization=True, **oi), f"{lensmodel} {core}{dist}{ext}{frames}{warp}{reg}")


def test_nothing_to_optimize_raises(amd, golden):
    kw = golden_case_inputs(golden, 0)
    for k in list(kw):
        if k.startswith("do_optimize"):
            kw[k] = False
    with pytest.raises(RuntimeError):
        amd.optimizer_callback(**kw)


def test_single_camera_odd_board(amd, ref_api):
    """monocular, 9x7 board (an odd corner count exercises the tile tail)"""
    oi, _ = make_calibration_problem(amd._api, Ncameras=1, Nframes=6, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=9, object_height_n=7, seed=11)
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                      ref_api.optimizer_callback(no_factorization=True, **oi))


def test_large_board(amd, ref_api):
    """more than 64 and more than 128 corners per board: several passes"""
    oi, _ = make_calibration_problem(amd._api, Ncameras=2, Nframes=3, lensmodel="LENSMODEL_OPENCV5",
                                     object_width_n=14, object_height_n=11, object_spacing=0.05, seed=13)
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                      ref_api.optimizer_callback(no_factorization=True, **oi))


def test_near_singular_rotations(amd, ref_api):
    """tiny / zero / near-pi rotations drive the special branches of the
    Rodrigues composition (poseutils-uses-autodiff.cc:455-768)"""
    oi, _ = make_calibration_problem(amd._api, Ncameras=2, Nframes=8, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=6, object_height_n=5, seed=17)
    oi["rt_ref_frame"][0,:3] = 0.
    oi["rt_ref_frame"][1,:3] = 1e-9
    oi["rt_ref_frame"][2,:3] = (1e-6, -2e-6, 5e-7)
    oi["rt_cam_ref"][0,:3]   = 0.
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                      ref_api.optimizer_callback(no_factorization=True, **oi), "camera r=0")
    oi["rt_cam_ref"][0,:3]   = (1e-9, 0, 1e-10)
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                      ref_api.optimizer_callback(no_factorization=True, **oi), "camera r tiny")
    # composition landing near a full turn: rc ~ (pi - small) about the same axis as rf
    oi["rt_cam_ref"][0,:3]   = (0, 0, 3.0)
    oi["rt_ref_frame"][3,:3] = (0, 0, 3.2)
    oi["rt_ref_frame"][4,:3] = (0, 0, 2*np.pi-3.0 - 1e-9)
    compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                      ref_api.optimizer_callback(no_factorization=True, **oi), "near 2pi")
