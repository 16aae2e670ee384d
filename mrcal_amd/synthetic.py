"""Synthetic chessboard-calibration problems: the workload generator for the
parity tests and the benchmark.

Restates, with numpy only, the behaviour of the reference's
mrcal.synthesize_board_observations() (mrcal/synthetic_data.py:236-554) and the
input preparation of test/test-basic-calibration.py:25-135 +
test/test_calibration_helpers.py:13-35:

  - stationary cameras (camera 0 at the reference), a board of WxH corners
    thrown around a nominal pose with uniform noise, only the poses where every
    camera sees the whole board are kept
  - per-corner weights uniform in [0.6,1], pixel noise N(0, sigma/weight), 1%
    of the corners pushed 20x further out (outliers to be found by the solver)
  - the solver is seeded from a perturbed copy of the truth

Perfect pixel observations need the lens model. Rather than carrying a second
projection implementation, the generator asks the library it is given: an
optimizer_callback() evaluation with all observed pixels at 0 and all weights at
1 returns x = q_hypothesis (mrcal.c:4712), which is what
make_perfect_observations() computes (mrcal/synthetic_data.py:594-757).
"""
import numpy as np

# The two cameras of the reference's test fixtures
# (test/data/cam0.opencv8.cameramodel, test/data/cam1.opencv8.cameramodel):
# fx fy cx cy k0..k7, imager 4000x2200. test-basic-calibration.py alternates them
CAM_OPENCV8 = (
    np.array((1761.181055, 1761.250444, 1965.706996, 1087.518797,
              -0.01266096516, 0.03590794372, -0.0002547045941, 0.0005275929652,
              0.01968883397, 0.01482863541, -0.0562239888, 0.0500223357)),
    np.array((1761., 1761., 1965., 1087.,
              -0.02, 0.03, 0.0002, 0.0005,
              0.0196, 0.01, -0.05, 0.04)))
IMAGERSIZE = (4000, 2200)


def R_from_r(r):
    """Rodrigues. r: (...,3) -> (...,3,3)"""
    r  = np.asarray(r, dtype=float)
    th = np.linalg.norm(r, axis=-1)[..., None, None]
    K  = np.zeros(r.shape[:-1] + (3,3))
    K[...,0,1] = -r[...,2]; K[...,0,2] =  r[...,1]
    K[...,1,0] =  r[...,2]; K[...,1,2] = -r[...,0]
    K[...,2,0] = -r[...,1]; K[...,2,1] =  r[...,0]
    small = th < 1e-8
    ths   = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(ths)/ths)
    b = np.where(small, 0.5, (1.0 - np.cos(ths))/(ths*ths))
    return np.eye(3) + a*K + b*(K @ K)


def intrinsics_for(lensmodel, Ncameras, seed=0):
    """Truth intrinsics for a lens model. The OPENCV family is cut down from
    the OPENCV8 fixture cameras as test-basic-calibration.py:38-41 does for
    OPENCV4; CAHVOR(E) and the splined models get small made-up distortions on
    the same cores"""
    import re
    rng = np.random.RandomState(1000 + seed)
    m = re.match(r"LENSMODEL_SPLINED_STEREOGRAPHIC_order=(\d+)_Nx=(\d+)_Ny=(\d+)_fov_x_deg=(\d+)", lensmodel)
    if m:
        Nx, Ny = int(m.group(2)), int(m.group(3))
        out = np.zeros((Ncameras, 4 + 2*Nx*Ny))
        for i in range(Ncameras):
            out[i,:4] = CAM_OPENCV8[i % 2][:4]
            # stereographic: q = 2 tan(th/2) f + c; keep the board inside the imager
            out[i,:2] *= 0.9
            out[i,4:] = rng.uniform(-0.01, 0.01, 2*Nx*Ny)
        return out
    if lensmodel.startswith("LENSMODEL_CAHVORE"):
        out = np.zeros((Ncameras, 12))
        for i in range(Ncameras):
            out[i,:4] = CAM_OPENCV8[i % 2][:4]
            out[i,4:] = (0.01, -0.02, 0.002, -0.05, 0.01, 0.002, 0.003, -0.001) + rng.uniform(-1e-3, 1e-3, 8)
        return out
    if lensmodel == "LENSMODEL_CAHVOR":
        out = np.zeros((Ncameras, 9))
        for i in range(Ncameras):
            out[i,:4] = CAM_OPENCV8[i % 2][:4]
            out[i,4:] = (0.01, -0.02, 0.002, -0.05, 0.01) + rng.uniform(-1e-3, 1e-3, 5)
        return out
    N = {"LENSMODEL_PINHOLE": 4, "LENSMODEL_STEREOGRAPHIC": 4,
         "LENSMODEL_LONLAT": 4, "LENSMODEL_LATLON": 4,
         "LENSMODEL_OPENCV4": 8, "LENSMODEL_OPENCV5": 9,
         "LENSMODEL_OPENCV8": 12, "LENSMODEL_OPENCV12": 16}[lensmodel]
    out = np.zeros((Ncameras, N))
    for i in range(Ncameras):
        src = CAM_OPENCV8[i % 2]
        n = min(N, 12)
        out[i,:n] = src[:n]
        if lensmodel in ("LENSMODEL_LONLAT", "LENSMODEL_LATLON"):
            # angular models: pixels per radian
            out[i,:2] = 1200.
    return out


def make_calibration_problem(api, *,
                             Ncameras, Nframes,
                             lensmodel         = "LENSMODEL_OPENCV8",
                             object_width_n    = 10,
                             object_height_n   = 10,
                             object_spacing    = 0.1,
                             calobject_warp    = (0.002, -0.005),
                             pixel_noise       = 1.5,
                             make_outliers     = True,
                             seed              = 0,
                             seed_perturbation = 1.0,
                             camera_spacing    = 0.3,
                             board_distance    = 4.0,
                             do_optimize_intrinsics_core = True):
    """Returns (optimization_inputs, truth). optimization_inputs is a dict
    ready for api.optimize(**optimization_inputs); all blocks optimized,
    regularization and outlier rejection on. api: a mrcal_amd._api.Api (any
    backing library). do_optimize_intrinsics_core=False is what
    mrcal-calibrate-cameras:638-643 does for the splined models (BASELINE.json's
    configuration 2)"""
    rng = np.random.RandomState(seed)
    W, H = object_width_n, object_height_n

    intrinsics_true = intrinsics_for(lensmodel, Ncameras)
    imagersizes     = np.array((IMAGERSIZE,)*Ncameras, dtype=np.int32)

    # cameras in a row along x, slightly mis-pointed; camera 0 is the reference
    rt_cam_ref_true = np.zeros((Ncameras-1, 6))
    for i in range(1, Ncameras):
        rt_cam_ref_true[i-1,:3] = rng.uniform(-0.05, 0.05, size=3)
        # rt_cam_ref transforms FROM the reference: a camera sitting at +x has
        # negative t_x
        rt_cam_ref_true[i-1,3:] = (-camera_spacing*i, rng.uniform(-0.05,0.05), rng.uniform(-0.05,0.05))
    x_center = camera_spacing*(Ncameras-1)/2.

    rt_ref_boardcenter = np.array((0., 0., 0., x_center, 0., board_distance))
    noiseradius = np.array((np.pi/180.*30., np.pi/180.*30., np.pi/180.*20., 2.5, 2.5, 2.0))
    board_center = np.array(((W-1)*object_spacing/2., (H-1)*object_spacing/2., 0.))
    warp_true = None if calobject_warp is None else np.array(calobject_warp, dtype=float)

    idx_cam = np.zeros((Ncameras, 3), dtype=np.int32)
    idx_cam[:,1] = np.arange(Ncameras)
    idx_cam[:,2] = np.arange(Ncameras) - 1

    def project_frames(rt_ref_frame):
        """perfect q of every corner for every (frame,camera): (Nf,Ncam,H,W,2)"""
        Nf = rt_ref_frame.shape[0]
        idx = np.tile(idx_cam, (Nf,1))
        idx[:,0] = np.repeat(np.arange(Nf, dtype=np.int32), Ncameras)
        obs = np.zeros((Nf*Ncameras, H, W, 3))
        obs[...,2] = 1.0
        x = api.optimizer_callback(
            intrinsics   = intrinsics_true,
            rt_cam_ref   = rt_cam_ref_true,
            rt_ref_frame = rt_ref_frame,
            observations_board = obs,
            indices_frame_camintrinsics_camextrinsics = idx,
            lensmodel    = lensmodel,
            imagersizes  = imagersizes,
            calobject_warp = warp_true,
            calibration_object_spacing = object_spacing,
            do_optimize_calobject_warp = False,
            do_apply_regularization    = False,
            no_jacobian = True, no_factorization = True)[1]
        return x[:Nf*Ncameras*H*W*2].reshape(Nf, Ncameras, H, W, 2)

    q_all  = np.zeros((0, Ncameras, H, W, 2))
    rt_all = np.zeros((0, 6))
    while q_all.shape[0] < Nframes:
        Nchunk = max(Nframes, 16)
        randomblock = rng.uniform(-1.0, 1.0, size=(Nchunk, 6))
        rt_center   = rt_ref_boardcenter + randomblock*noiseradius
        # board-corner-referenced pose: same rotation, origin moved to the corner
        rt_frame        = rt_center.copy()
        rt_frame[:,3:] -= (R_from_r(rt_center[:,:3]) @ board_center[:,None])[...,0]
        q = project_frames(rt_frame)
        visible = (q[...,0] >= 0) & (q[...,1] >= 0) & \
                  (q[...,0] <= IMAGERSIZE[0]-1) & (q[...,1] <= IMAGERSIZE[1]-1) & \
                  np.isfinite(q[...,0]) & np.isfinite(q[...,1])
        # all cameras must see the full board, and the board must face them
        keep = np.all(visible, axis=(1,2,3))
        q_all  = np.concatenate((q_all,  q[keep]))
        rt_all = np.concatenate((rt_all, rt_frame[keep]))
    q_all  = q_all [:Nframes]
    rt_ref_frame_true = rt_all[:Nframes]

    weight = 0.2 + 0.8*(rng.rand(Nframes, Ncameras, H, W) + 1.)/2.
    q_noise = rng.randn(Nframes, Ncameras, H, W, 2) * pixel_noise / weight[...,None]
    if make_outliers:
        Npts = Nframes*Ncameras*H*W
        i_outliers = rng.choice(Npts, (Npts//100,), replace=False)
        q_noise.reshape(Npts,2)[i_outliers] *= 20.
    observations = np.concatenate((q_all + q_noise, weight[...,None]), axis=-1) \
                     .reshape(Nframes*Ncameras, H, W, 3)

    indices = np.tile(idx_cam, (Nframes,1))
    indices[:,0] = np.repeat(np.arange(Nframes, dtype=np.int32), Ncameras)

    # seed: the truth, perturbed
    s = seed_perturbation
    intrinsics = intrinsics_true.copy()
    intrinsics[:,:2] *= 1. + s*0.01*rng.uniform(-1,1,size=(Ncameras,2))
    intrinsics[:,2:4] += s*5.*rng.uniform(-1,1,size=(Ncameras,2))
    if intrinsics.shape[1] > 4:
        intrinsics[:,4:] *= 1. + s*0.1*rng.uniform(-1,1,size=intrinsics[:,4:].shape)
    rt_cam_ref = rt_cam_ref_true + s*rng.uniform(-1,1,size=rt_cam_ref_true.shape) * \
        np.array((2e-3,2e-3,2e-3, 1e-2,1e-2,1e-2))
    rt_ref_frame = rt_ref_frame_true + s*rng.uniform(-1,1,size=rt_ref_frame_true.shape) * \
        np.array((5e-3,5e-3,5e-3, 2e-2,2e-2,2e-2))

    optimization_inputs = dict(
        intrinsics   = np.ascontiguousarray(intrinsics),
        rt_cam_ref   = np.ascontiguousarray(rt_cam_ref),
        rt_ref_frame = np.ascontiguousarray(rt_ref_frame),
        points       = None,
        observations_board = np.ascontiguousarray(observations),
        indices_frame_camintrinsics_camextrinsics = np.ascontiguousarray(indices),
        observations_point = None,
        indices_point_camintrinsics_camextrinsics = None,
        lensmodel    = lensmodel,
        imagersizes  = imagersizes,
        calobject_warp = None if warp_true is None else np.zeros((2,)),
        calibration_object_spacing = object_spacing,
        do_optimize_intrinsics_core        = bool(do_optimize_intrinsics_core),
        do_optimize_intrinsics_distortions = intrinsics.shape[1] > 4,
        do_optimize_extrinsics             = Ncameras > 1,
        do_optimize_frames                 = True,
        do_optimize_calobject_warp         = warp_true is not None,
        do_apply_regularization            = True,
        do_apply_outlier_rejection         = True,
        verbose                            = False)
    truth = dict(intrinsics   = intrinsics_true,
                 rt_cam_ref   = rt_cam_ref_true,
                 rt_ref_frame = rt_ref_frame_true,
                 calobject_warp = warp_true,
                 q            = q_all)
    return optimization_inputs, truth


def copy_inputs(optimization_inputs):
    """deep copy of the arrays: optimize() works in place"""
    return { k: (v.copy() if isinstance(v, np.ndarray) else v)
             for k,v in optimization_inputs.items() }


def _rodrigues_one(r):
    """Rodrigues, one vector (the SfM generator's own: kept as it was so that its problems keep their bits)"""
    th = np.linalg.norm(r)
    if th < 1e-12: return np.eye(3)
    k = r/th
    K = np.array(((0,-k[2],k[1]),(k[2],0,-k[0]),(-k[1],k[0],0)))
    return np.eye(3) + np.sin(th)*K + (1-np.cos(th))*(K@K)


def make_sfm_problem(lensmodel="LENSMODEL_PINHOLE", Ncam=4, Npoints=40, seed=0, noise=0.5, Nboard_frames=0,
                board_wh=(10,10), board_spacing=0.1):
    """An SfM-shaped problem (BASELINE.json's configurations 4 and 5, test/test-sfm-triangulated-points.py's shape):
    Ncam cameras in a row, Npoints points seen by 2..Ncam of them each as triangulated observations with pixel noise,
    intrinsics locked, extrinsics optimized from a perturbed seed, unity_cam01 regularization. Returns
    (optimization_inputs, truth).
    Nboard_frames > 0: chessboard frames beside the triangulated points, every camera seeing every board
    (BASELINE.json's configuration 5: "4 cameras + 20k triangulated points + board frames"; allowed by
    mrcal.c:6043-6051 with the intrinsics locked). The frame poses are then optimized too"""
    rng = np.random.RandomState(seed)
    W, H = 4000, 2200
    core = np.array((600., 600., (W-1)/2., (H-1)/2.))
    if lensmodel == "LENSMODEL_PINHOLE":
        intr = np.tile(core, (Ncam,1))
    elif lensmodel == "LENSMODEL_OPENCV4":
        intr = np.tile(np.r_[core, -0.01, 0.02, 1e-3, -2e-3], (Ncam,1))
    else:
        raise ValueError(lensmodel)
    # camera 0 at the reference, the others ~1m apart along x, slightly rotated
    rt_cam_ref = np.zeros((Ncam-1, 6))
    for i in range(1, Ncam):
        rt_cam_ref[i-1,:3] = rng.uniform(-0.05, 0.05, 3)
        rt_cam_ref[i-1,3:] = (-1.0*i, rng.uniform(-0.1,0.1), rng.uniform(-0.1,0.1))

    def pixel(ic, pref):
        """perfect pixel of a point given in the reference frame, seen by camera ic"""
        p = pref if ic == 0 else _rodrigues_one(rt_cam_ref[ic-1,:3]) @ pref + rt_cam_ref[ic-1,3:]
        x, y = p[0]/p[2], p[1]/p[2]
        if lensmodel == "LENSMODEL_OPENCV4":
            k = intr[ic,4:]
            r2 = x*x + y*y
            cd = 1 + k[0]*r2 + k[1]*r2*r2
            x, y = x*cd + 2*k[2]*x*y + k[3]*(r2+2*x*x), y*cd + k[2]*(r2+2*y*y) + 2*k[3]*x*y
        return core[:2]*np.array((x,y)) + core[2:]

    pts = np.column_stack((rng.uniform(-3, 5, Npoints), rng.uniform(-2, 2, Npoints), rng.uniform(8, 30, Npoints)))
    obs, idx = [], []
    for ip in range(Npoints):
        cams = np.sort(rng.choice(Ncam, size=rng.randint(2, Ncam+1), replace=False))
        for ic in cams:
            q = pixel(ic, pts[ip]) + rng.normal(0, noise, 2)
            obs.append((q[0], q[1], 1.0))
            idx.append((ip, ic, ic-1))
    obs = np.array(obs); idx = np.array(idx, dtype=np.int32)
    obs[5,2] = -1.      # an outlier on input
    # seed: the truth, perturbed
    seedrt = rt_cam_ref + rng.normal(0, 1, rt_cam_ref.shape)*np.array((0.01,0.01,0.01,0.05,0.05,0.05))
    oi = dict(intrinsics = intr, lensmodel = lensmodel,
              imagersizes = np.tile(np.array((W,H), dtype=np.int32), (Ncam,1)),
              rt_cam_ref = np.ascontiguousarray(seedrt),
              observations_point_triangulated = np.ascontiguousarray(obs),
              indices_point_triangulated_camintrinsics_camextrinsics = np.ascontiguousarray(idx),
              do_optimize_intrinsics_core = False, do_optimize_intrinsics_distortions = False,
              do_optimize_extrinsics = True, do_optimize_frames = False, do_optimize_calobject_warp = False,
              do_apply_regularization = True, do_apply_regularization_unity_cam01 = True,
              do_apply_outlier_rejection = False, verbose = False)
    truth = dict(rt_cam_ref=rt_cam_ref, points=pts)
    if Nboard_frames:
        Wb, Hb = board_wh
        rb = np.random.RandomState(seed + 77771)     # (its own stream: the points above do not depend on the boards)
        rt_ref_frame = np.column_stack((rb.uniform(-0.3, 0.3, (Nboard_frames,3)),
                                        rb.uniform(-2.5, 0.5, Nboard_frames), rb.uniform(-1, 0.5, Nboard_frames),
                                        rb.uniform(4, 8, Nboard_frames)))
        gx, gy = np.meshgrid(np.arange(Wb)*board_spacing, np.arange(Hb)*board_spacing)
        corners = np.stack((gx, gy, np.zeros_like(gx)), axis=-1)          # (Hb,Wb,3): y-major then x (mrcal.c:2794)
        ob = np.zeros((Nboard_frames*Ncam, Hb, Wb, 3))
        ib = np.zeros((Nboard_frames*Ncam, 3), dtype=np.int32)
        for f in range(Nboard_frames):
            pref = corners @ _rodrigues_one(rt_ref_frame[f,:3]).T + rt_ref_frame[f,3:]
            for ic in range(Ncam):
                o = f*Ncam + ic
                ib[o] = (f, ic, ic-1)
                for iy in range(Hb):
                    for ix in range(Wb):
                        ob[o,iy,ix,:2] = pixel(ic, pref[iy,ix])
        ob[...,:2] += rb.normal(0, noise, ob[...,:2].shape)
        ob[...,2]   = rb.uniform(0.5, 1.0, ob.shape[:3])
        ob[1, min(2, Hb-1), min(3, Wb-1), 2] = -1.      # an outlier on input
        oi.update(rt_ref_frame = np.ascontiguousarray(rt_ref_frame + rb.normal(0, 1, rt_ref_frame.shape)*np.array((5e-3,)*3 + (2e-2,)*3)),
                  observations_board = ob, indices_frame_camintrinsics_camextrinsics = ib,
                  calibration_object_spacing = board_spacing, do_optimize_frames = True)
        truth["rt_ref_frame"] = rt_ref_frame
    return oi, truth


# BASELINE.json's configuration 2 as SURVEY.md section 8(d) wrote it: 30 x 20 control points over 150 degrees
CONFIG2_LENSMODEL = "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=30_Ny=20_fov_x_deg=150"
