"""Seeding: the step BEFORE the optimizer. From chessboard observations alone,
a rough stereographic model per camera, the cameras' poses relative to camera 0
and the board's pose in every frame, good enough for optimize() to start from
(reference: mrcal/calibration.py:508-1610,
traverse-sensor-links.c, called from mrcal-calibrate-cameras:412-535).

Same entry points and conventions as the reference:

    intrinsics_core, rt_cam_ref, rt_ref_frame = \\
        seed_stereographic(imagersizes, focal_estimate, indices_frame_camera, observations, object_spacing)
    Rt_cam_frame = estimate_monocular_calobject_poses_Rt_tocam(indices_frame_camera, observations,
                                                               object_spacing, models_or_intrinsics)
    rt_ref_frame = estimate_joint_frame_poses(Rt_cam_frame, Rt_cam_ref, indices_frame_camera, W, H, object_spacing)

What is different is HOW the per-observation pose comes about. The reference
re-projects every observation to a pinhole image and hands it to OpenCV's
solvePnP, one observation at a time, retrying with other focal lengths when a
wide lens puts corners near or behind the image plane of that pinhole (its own
comment calls this a hack and asks for a solver that works on observation
VECTORS). That is what is done here, for all observations at once:

  1. all corners of all observations are unprojected in ONE batch on the GPU
     (mrcal_amd.unproject: any lens model, the same device code the solver uses)
  2. planar pose from vectors, batched over the observations (numpy): the 3x3
     matrix M = [r1 r2 t] with v_i x (M [X_i Y_i 1]^T) = 0 from the smallest
     singular vector, its scale and sign fixed by |r1| = |r2| = 1 and "the
     board is where the camera looks", r3 = r1 x r2, nearest rotation
  3. a few Gauss-Newton steps on the angular error sum |v_i/|v_i| - p_i/|p_i||^2
     over (r, t), batched the same way

No pinhole image is involved, so there is nothing to retry: corners at 90
degrees off axis or beyond are as good as any other.
"""
import heapq

import numpy as np

from . import poseutils as _pu


def ref_calibration_object(W=None, H=None, object_spacing=None, *, calobject_warp=None, optimization_inputs=None):
    """corner positions of the board in its own coordinates, (H,W,3): x along the
    width, y along the height, z = the parabolic deflection the solver models
    (mrcal.c:2794-2830: x2, y2 = the deflection at the middle of the x and y
    extents). reference: mrcal/synthetic_data.py ref_calibration_object"""
    if optimization_inputs is not None:
        if not (W is None and H is None and object_spacing is None):
            raise Exception("W,H,object_spacing and optimization_inputs cannot both be given")
        H, W = optimization_inputs["observations_board"].shape[-3:-1]
        object_spacing = optimization_inputs["calibration_object_spacing"]
        calobject_warp = optimization_inputs["calobject_warp"]
    elif W is None or H is None or object_spacing is None:
        raise Exception("W,H,object_spacing: ALL must be given, or optimization_inputs")
    xx, yy = np.meshgrid(np.arange(W, dtype=float), np.arange(H, dtype=float))
    obj = np.stack((xx*object_spacing, yy*object_spacing, np.zeros_like(xx)), axis=-1)
    if calobject_warp is not None:
        xr = xx/(W - 1.)
        yr = yy/(H - 1.)
        obj[..., 2] += calobject_warp[0]*4.*xr*(1. - xr) + calobject_warp[1]*4.*yr*(1. - yr)
    return obj


def align_procrustes_points_Rt01(p0, p1, weights=None):
    """the rigid transform with p0 ~ R p1 + t in the least-squares sense, (4,3)
    (Kabsch; reference: mrcal/poseutils.py align_procrustes_points_Rt01).
    p0, p1: (N,3)"""
    p0 = np.asarray(p0, dtype=float).reshape(-1, 3)
    p1 = np.asarray(p1, dtype=float).reshape(-1, 3)
    if p0.shape != p1.shape or p0.shape[0] < 3:
        raise Exception("align_procrustes_points_Rt01() needs two matching sets of at least 3 points")
    w = np.ones(p0.shape[0]) if weights is None else np.asarray(weights, dtype=float)
    c0 = (w[:, None]*p0).sum(0)/w.sum()
    c1 = (w[:, None]*p1).sum(0)/w.sum()
    M = ((p0 - c0)*w[:, None]).T @ (p1 - c1)
    U, s, Vt = np.linalg.svd(M)
    if s[1] < 1e-12*max(s[0], 1e-300):
        raise Exception("align_procrustes_points_Rt01(): the points are (nearly) collinear; the rotation is not determined")
    D = np.diag((1., 1., np.sign(np.linalg.det(U @ Vt))))
    R = U @ D @ Vt
    return np.vstack((R, c0 - R @ c1))


def traverse_sensor_links(*, connectivity_matrix, callback_sensor_link):
    """Visits every sensor reachable from sensor 0 in the order of its distance
    from sensor 0 and reports the edge it was best reached through:
    callback_sensor_link(idx_to, idx_from). An edge between two sensors that
    share n > 0 frames costs 65536 - n: fewest hops first, most shared frames
    among those (traverse-sensor-links.c:38-50). Unreachable sensors are not
    visited; that is for the caller to notice"""
    C = np.asarray(connectivity_matrix)
    N = C.shape[0]
    if C.shape != (N, N) or np.any(C != C.T):
        raise Exception("the connectivity matrix must be square and symmetric")
    cost = [None]*N
    parent = [-1]*N
    done = [False]*N
    cost[0] = 0
    heap = [(0, 0)]
    while heap:
        c, i = heapq.heappop(heap)
        if done[i] or c != cost[i]: continue              # a stale entry
        done[i] = True
        if i != 0: callback_sensor_link(i, parent[i])
        for j in range(N):
            if j == i or done[j] or C[i, j] <= 0: continue
            cj = c + 65536 - int(C[i, j])
            if cost[j] is None or cj < cost[j]:
                cost[j], parent[j] = cj, i
                heapq.heappush(heap, (cj, j))


# ----------------------------------------------------------------------------- planar pose from vectors
def _planar_pose_from_vectors(v, XY, mask, Niterations=8):
    """v: (N,P,3) observation vectors in the camera frame (any length), XY:
    (P,2) the board points (z = 0), mask: (N,P) which of them count. Returns Rt
    (N,4,3) camera <- board and the rms angular error (N,) in radians. An
    observation with fewer than 4 usable points raises"""
    v = np.asarray(v, dtype=float)
    N, P = v.shape[:2]
    w = np.asarray(mask, dtype=float)
    if np.any(w.sum(-1) < 4):
        bad = np.nonzero(w.sum(-1) < 4)[0]
        raise Exception(f"Insufficient observations; need at least 4 corners; observations {bad.tolist()} have fewer")
    vn = v/np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-300)
    vn = np.where(w[..., None] > 0, vn, 0.)
    # centred, scaled board coordinates condition the linear system
    c = XY.mean(0)
    s = np.sqrt(((XY - c)**2).sum(-1).mean())
    Xh = np.concatenate(((XY - c)/s, np.ones((P, 1))), axis=-1)           # (P,3)

    # v x (M Xh) = 0: rows [v]x kron Xh^T, M row-major (m = M.ravel())
    vx, vy, vz = vn[..., 0], vn[..., 1], vn[..., 2]
    Z = np.zeros_like(vx)
    skew = np.stack((np.stack((Z, -vz, vy), -1), np.stack((vz, Z, -vx), -1), np.stack((-vy, vx, Z), -1)), -2)   # (N,P,3,3)
    A = (skew[..., :, :, None]*Xh[None, :, None, None, :]).reshape(N, P*3, 9)
    # the 9x9 normal matrix is enough: smallest eigenvector
    AtA = np.einsum("nri,nrj->nij", A, A)
    evals, evecs = np.linalg.eigh(AtA)
    M = evecs[..., 0].reshape(N, 3, 3)
    # undo the conditioning: M Xh = M T [X Y 1], T = [[1/s,0,-cx/s],[0,1/s,-cy/s],[0,0,1]]
    T = np.array(((1./s, 0., -c[0]/s), (0., 1./s, -c[1]/s), (0., 0., 1.)))
    M = M @ T
    # sign: the points are in front of the vectors that see them
    p = np.einsum("nij,pj->npi", M, np.concatenate((XY, np.ones((P, 1))), -1))
    sign = np.sign((np.einsum("npi,npi->np", p, vn)*w).sum(-1))
    sign = np.where(sign == 0, 1., sign)
    M = M*sign[:, None, None]
    scale = 2./(np.linalg.norm(M[..., 0], axis=-1) + np.linalg.norm(M[..., 1], axis=-1))
    M = M*scale[:, None, None]
    r1, r2, t = M[..., 0], M[..., 1], M[..., 2]
    R0 = np.stack((r1, r2, np.cross(r1, r2)), axis=-1)
    U, _, Vt = np.linalg.svd(R0)
    det = np.linalg.det(U @ Vt)
    U[..., :, 2] *= det[:, None]
    R = U @ Vt

    # Gauss-Newton on e_i = vn_i - p_i/|p_i|, p_i = R X_i + t; the rotation is
    # updated on the left: R <- R(dr) R
    X3 = np.concatenate((XY, np.zeros((P, 1))), -1)
    def residual(R, t):
        p = np.einsum("nij,pj->npi", R, X3) + t[:, None, :]
        d = np.maximum(np.linalg.norm(p, axis=-1, keepdims=True), 1e-300)
        return p, d, (vn - p/d)*w[..., None]
    p, d, e = residual(R, t)
    cost = (e*e).sum((-1, -2))
    lam = np.zeros(N)
    for _ in range(Niterations):
        u = p/d
        # d(p/|p|)/dp = (I - u u^T)/|p|
        Pj = (np.eye(3) - u[..., :, None]*u[..., None, :])/d[..., None]                 # (N,P,3,3)
        Rx = p - t[:, None, :]                                                              # R X_i
        Zr = np.zeros_like(Rx[..., 0])
        # dp/d(dr) = -[R X]x ; dp/dt = I
        Kx = np.stack((np.stack((Zr, Rx[..., 2], -Rx[..., 1]), -1),
                       np.stack((-Rx[..., 2], Zr, Rx[..., 0]), -1),
                       np.stack((Rx[..., 1], -Rx[..., 0], Zr), -1)), -2)                   # -[R X]x
        J = -np.concatenate((Pj @ Kx, Pj), axis=-1)*w[..., None, None]                     # de/d(dr,dt): (N,P,3,6)
        JtJ = np.einsum("npki,npkj->nij", J, J)
        Jte = np.einsum("npki,npk->ni", J, e)
        JtJ = JtJ + (lam[:, None, None] + 1e-12)*np.eye(6)*np.trace(JtJ, axis1=-2, axis2=-1)[:, None, None]
        step = -np.linalg.solve(JtJ, Jte[..., None])[..., 0]
        Rn = _pu.R_from_r(step[:, :3]) @ R
        tn = t + step[:, 3:]
        pn, dn, en = residual(Rn, tn)
        costn = (en*en).sum((-1, -2))
        better = costn <= cost
        R = np.where(better[:, None, None], Rn, R)
        t = np.where(better[:, None], tn, t)
        p = np.where(better[:, None, None], pn, p)
        d = np.where(better[:, None, None], dn, d)
        e = np.where(better[:, None, None], en, e)
        cost = np.where(better, costn, cost)
        lam = np.where(better, lam*0.1, np.maximum(lam, 1e-4)*10.)
    rms = np.sqrt(cost/np.maximum(w.sum(-1), 1.))
    return np.concatenate((R, t[:, None, :]), axis=-2), rms


def _intrinsics_of(models_or_intrinsics):
    out = []
    for m in models_or_intrinsics:
        li = m.intrinsics() if hasattr(m, "intrinsics") and callable(m.intrinsics) else m
        out.append((str(li[0]), np.asarray(li[1], dtype=float)))
    return out


def estimate_monocular_calobject_poses_Rt_tocam(indices_frame_camera, observations, object_spacing,
                                                models_or_intrinsics, *, paths=None):
    """Pose of the board in the camera that sees it, for every observation
    separately: (Nobservations,4,3), camera <- board (calibration.py:622-780).
    observations: (Nobservations,H,W,3) rows (x, y, weight); corners with a
    negative x, y or weight are ignored. models_or_intrinsics: per camera, a
    cameramodel or a (lensmodel, intrinsics_data) pair"""
    from . import unproject
    indices_frame_camera = np.asarray(indices_frame_camera)
    observations = np.asarray(observations, dtype=float)
    Nobs, H, W = observations.shape[:3]
    intr = _intrinsics_of(models_or_intrinsics)
    XY = ref_calibration_object(W, H, object_spacing).reshape(-1, 3)[:, :2]
    q = np.ascontiguousarray(observations[..., :2].reshape(Nobs, H*W, 2))
    mask = (observations[..., 2].reshape(Nobs, H*W) > 0) & (q[..., 0] >= 0) & (q[..., 1] >= 0)
    v = np.zeros((Nobs, H*W, 3))
    for icam, (lensmodel, data) in enumerate(intr):
        sel = np.nonzero(indices_frame_camera[:, 1] == icam)[0]
        if sel.size == 0: continue
        # masked-out corners may hold anything (-1,-1 by convention): give the lens model something harmless
        qq = np.where(mask[sel][..., None], q[sel], data[2:4])
        v[sel] = unproject(np.ascontiguousarray(qq), lensmodel, data)
    mask &= np.all(np.isfinite(v), axis=-1)
    v = np.where(mask[..., None], v, 0.)
    few = np.nonzero(mask.sum(-1) < 4)[0]
    if few.size:
        i = int(few[0])
        what = f"observation {i} (camera {int(indices_frame_camera[i,1])}" + (f'; "{paths[i]}"' if paths is not None else "") + ")"
        raise Exception(f"Insufficient observations; need at least 4; got {int(mask[i].sum())} instead. "
                        f"Cannot estimate initial extrinsics for {what}")
    Rt, _ = _planar_pose_from_vectors(v, XY, mask)
    return Rt


def _estimate_camera_poses(calobject_poses_local_Rt_cf, indices_frame_camera, object_width_n, object_height_n, object_spacing):
    """camera i -> camera 0 for i = 1..Ncameras-1, (Ncameras-1,4,3), from the
    frames that pairs of cameras see together, chained along the best-connected
    path to camera 0 (calibration.py:925-1100)"""
    idx = np.asarray(indices_frame_camera)
    Ncameras = int(idx[:, 1].max()) + 1
    if np.any(np.diff(idx[:, 0]) < 0):
        raise Exception("I'm assuming the frame indices are increasing monotonically")
    obj = ref_calibration_object(object_width_n, object_height_n, object_spacing).reshape(-1, 3)
    # which observation is (frame, camera)
    frames, inv = np.unique(idx[:, 0], return_inverse=True)
    table = -np.ones((frames.size, Ncameras), dtype=int)
    for i, (f, c) in enumerate(zip(inv, idx[:, 1])):
        if table[f, c] >= 0: raise Exception(f"Saw multiple camera{c} observations in frame {idx[i,0]}")
        table[f, c] = i
    seen = table >= 0
    shared = (seen[:, :, None] & seen[:, None, :]).sum(0)
    np.fill_diagonal(shared, 0)
    shared[shared < 2] = 0                      # one shared frame does not pin a relative pose down well enough

    def pairwise_Rt(icam_to, icam_from):
        both = np.nonzero(seen[:, icam_to] & seen[:, icam_from])[0]
        A = _pu.transform_point_Rt(calobject_poses_local_Rt_cf[table[both, icam_to]][:, None], obj[None])
        B = _pu.transform_point_Rt(calobject_poses_local_Rt_cf[table[both, icam_from]][:, None], obj[None])
        return align_procrustes_points_Rt01(A.reshape(-1, 3), B.reshape(-1, 3))

    Rt_0c = [None]*(Ncameras - 1)
    def link(icam, ifrom):
        Rt_fc = pairwise_Rt(ifrom, icam)
        Rt_0c[icam - 1] = Rt_fc if ifrom == 0 else _pu.compose_Rt(Rt_0c[ifrom - 1], Rt_fc)
    traverse_sensor_links(connectivity_matrix=shared, callback_sensor_link=link)
    if any(x is None for x in Rt_0c):
        raise Exception("ERROR: Don't have complete camera observations overlap!\n"
                        f"Shared observations matrix:\n{shared}\n")
    return np.array(Rt_0c).reshape(-1, 4, 3)


def estimate_joint_frame_poses(calobject_Rt_camera_frame, Rt_cam_ref, indices_frame_camera,
                               object_width_n, object_height_n, object_spacing):
    """Pose of the board in the reference (camera 0) frame for every frame,
    (Nframes,6) rt reference <- board. A frame seen by several cameras: the
    board fitted to the mean of the point clouds the cameras put it at
    (calibration.py:1186-1395)"""
    idx = np.asarray(indices_frame_camera)
    Rt_cf = np.asarray(calobject_Rt_camera_frame, dtype=float)
    Rt_ref_cam = _pu.invert_Rt(np.asarray(Rt_cam_ref, dtype=float).reshape(-1, 4, 3))
    obj = ref_calibration_object(object_width_n, object_height_n, object_spacing).reshape(-1, 3)
    # every observation's board pose in the reference frame
    Rt_rf = np.array(Rt_cf)
    other = idx[:, 1] > 0
    if np.any(other):
        Rt_rf[other] = _pu.compose_Rt(Rt_ref_cam[idx[other, 1] - 1], Rt_cf[other])
    out = []
    starts = np.concatenate(((0,), np.nonzero(np.diff(idx[:, 0]) != 0)[0] + 1, (idx.shape[0],)))
    for i0, i1 in zip(starts[:-1], starts[1:]):
        if i1 - i0 == 1:
            out.append(_pu.rt_from_Rt(Rt_rf[i0]))
        else:
            mean = _pu.transform_point_Rt(Rt_rf[i0:i1, None], obj[None]).mean(0)
            out.append(_pu.rt_from_Rt(align_procrustes_points_Rt01(mean, obj)))
    return np.array(out).reshape(-1, 6)


def seed_stereographic(imagersizes, focal_estimate, indices_frame_camera, observations, object_spacing, *, paths=None):
    """(intrinsics_core (Ncameras,4), rt_cam_ref (Ncameras-1,6), rt_ref_frame
    (Nframes,6)): a LENSMODEL_STEREOGRAPHIC camera per imager with the given
    focal length and the centre of the imager as its centre pixel, and the
    geometry that goes with it (calibration.py:1398-1608). To calibrate a richer
    model, pad the core with zeros and optimize in stages, as
    mrcal-calibrate-cameras does"""
    Ncameras = len(imagersizes)
    try:    focal = list(focal_estimate)
    except TypeError: focal = [focal_estimate]
    if len(focal) == 1: focal = focal*Ncameras
    if len(focal) != Ncameras:
        raise Exception(f"Ncameras mismatch: len(imagersizes) = {Ncameras} but len(focal_estimate) = {len(focal)}")
    intrinsics = [("LENSMODEL_STEREOGRAPHIC",
                   np.array((focal[i], focal[i], (imagersizes[i][0] - 1.)/2., (imagersizes[i][1] - 1.)/2.)))
                  for i in range(Ncameras)]
    Rt_cf = estimate_monocular_calobject_poses_Rt_tocam(indices_frame_camera, observations, object_spacing,
                                                        intrinsics, paths=paths)
    H, W = observations.shape[-3:-1]
    Rt_0c = _estimate_camera_poses(Rt_cf, indices_frame_camera, W, H, object_spacing)
    Rt_cam_ref = _pu.invert_Rt(Rt_0c) if len(Rt_0c) else np.zeros((0, 4, 3))
    rt_ref_frame = estimate_joint_frame_poses(Rt_cf, Rt_cam_ref, indices_frame_camera, W, H, object_spacing)
    return (np.array([i[1] for i in intrinsics]),
            _pu.rt_from_Rt(Rt_cam_ref).reshape(-1, 6),
            rt_ref_frame)
