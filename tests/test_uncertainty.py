"""_mrcal_drt_cross_reprojection__dbpacked() / mrcal.drt_cross_reprojection__dbpacked() (SURVEY 8 f1;
uncertainty.c:798-1541, mrcal.h:613-669, mrcal-pywrap.c:2012-2110): K_packed = drt_ref_refperturbed/db_packed
("rrp") or drt_cam_camperturbed/db_packed ("ccp").

The checker is the reference's OWN uncertainty.c, compiled in place into oracle/_ref/libmrcal_ref.so (it needs
from SuiteSparse only the cholmod_sparse type, and from LAPACK the packed 6x6 Cholesky pair dpptrf/dpptrs, for
which oracle/lapack_packed_stub.c stands in). Pinned here, without a GPU, the way the reference's
test/test-projection-uncertainty.py:1960-2161 pins it: against the dense expression
K = -lstsq(J_cross, J_packed[e,f,p,cw]) in numpy, eps 1e-12 worst case.

GPU: the product (csrc/uncertainty.hip: the sums over J's rows on the device) against that checker on the same
inputs through the same ctypes binding, against the dense expression, and the resident tier against the
drop-in tier."""
import ctypes as C
import numpy as np
import pytest

from conftest import REFLIB_PATH
from mrcal_amd.synthetic import copy_inputs

SR_CAM, ST_CAM = 0.1*np.pi/180., 1.0      # scales.h:40-48
SR_FRM, ST_FRM = 15.0*np.pi/180., 1.0
S_POINT        = 1.0


def skew(t):
    return np.array(((0, -t[2], t[1]), (t[2], 0, -t[0]), (-t[1], t[0], 0)))


def compose_rt_tinyrt0_gradientrt0(rt):
    """d compose_rt(rt0, rt)/d rt0 at rt0 = 0: [[dr/dr0, 0], [-skew(t), I]] with dr/dr0 of
    poseutils.c:1003-1072 (B = |r|/2): B/tanB I - (B/tanB - 1)/(4 B^2) r r^T - skew(r)/2"""
    r, t = rt[:3], rt[3:]
    n2 = r @ r
    if n2 < 4e-16:
        M = np.eye(3)
    else:
        B = np.sqrt(n2)/2.
        c = B/np.tan(B)
        M = c*np.eye(3) - np.outer(r, r)*(c - 1.)/(4.*B*B) - skew(r)/2.
    T = np.zeros((6,6))
    T[:3,:3] = M
    T[3:,:3] = -skew(t)
    T[3:,3:] = np.eye(3)
    return T


def test_tinyrt0_gradient_is_the_derivative():
    """the closed form above against central differences of the numpy compose_rt (mrcal_amd/poseutils.py)"""
    from mrcal_amd.poseutils import compose_rt
    rng = np.random.RandomState(0)
    for _ in range(5):
        rt = np.r_[rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3)]
        T = compose_rt_tinyrt0_gradientrt0(rt)
        h = 1e-6
        Tn = np.zeros((6,6))
        for k in range(6):
            d = np.zeros(6); d[k] = h
            Tn[:,k] = (compose_rt(d, rt) - compose_rt(-d, rt))/(2*h)
        assert np.abs(T - Tn).max() < 1e-8


def dense_K(api, oi, icam):
    """-lstsq(J_cross, J_packed[efpcw]) as test-projection-uncertainty.py:1960-2161 builds it"""
    b, x, J, _ = api.optimizer_callback(no_factorization=True, **oi)
    Nstate = J.shape[1]
    Nmeas_obs = api.num_measurements_boards(**oi) + api.num_measurements_points(**oi)
    Jd = J[:Nmeas_obs].toarray()
    bu = b.copy(); api.unpack_state(bu, **oi)
    i_i, i_e, i_f, i_p, i_cw = (api.state_index_intrinsics(0, **oi), api.state_index_extrinsics(0, **oi),
                                api.state_index_frames(0, **oi), api.state_index_points(0, **oi),
                                api.state_index_calobject_warp(**oi))
    N_i, N_e, N_f, N_p = (api.num_states_intrinsics(**oi), api.num_states_extrinsics(**oi),
                          api.num_states_frames(**oi), api.num_states_points(**oi))
    Jc_fp = np.zeros((Nmeas_obs, 6))
    if i_f is not None and N_f:
        D = np.zeros((N_f, 6))
        for f in range(N_f//6):
            D[6*f:6*f+6] = compose_rt_tinyrt0_gradientrt0(bu[i_f+6*f:i_f+6*f+6]) / np.array((SR_FRM,)*3 + (ST_FRM,)*3)[:,None]
        Jc_fp += Jd[:, i_f:i_f+N_f] @ D
    if i_p is not None and N_p:
        D = np.zeros((N_p, 6))
        for k in range(N_p//3):
            p = bu[i_p+3*k:i_p+3*k+3]
            D[3*k:3*k+3, :3] = -skew(p)/S_POINT
            D[3*k:3*k+3, 3:] = np.eye(3)/S_POINT
        Jc_fp += Jd[:, i_p:i_p+N_p] @ D
    keep = np.ones(Nstate, dtype=bool)
    if i_i is not None and N_i: keep[i_i:i_i+N_i] = False
    if icam is None or icam < 0:
        if i_e is not None and N_e: keep[i_e:i_e+N_e] = False
        Jc = Jc_fp
    else:
        Jc = Jc_fp.copy()
        if i_e is not None and N_e:
            D = np.zeros((N_e, 6))
            for e in range(N_e//6):
                D[6*e:6*e+6] = compose_rt_tinyrt0_gradientrt0(bu[i_e+6*e:i_e+6*e+6]) / np.array((SR_CAM,)*3 + (ST_CAM,)*3)[:,None]
            has_e = np.abs(Jd[:, i_e:i_e+N_e]).sum(axis=1) != 0
            # (a row whose camera has extrinsics carries them; an outlier's all-zero row contributes 0 either way)
            Jc_e = Jd[:, i_e:i_e+N_e] @ D
            Jc[has_e] = Jc_e[has_e]
        # rows of the other cameras do not count
        cam_of_row = np.full(Nmeas_obs, -1)
        if N_i:
            Nper = N_i // oi["intrinsics"].shape[0]
            first = np.array(J[:Nmeas_obs].indices[J.indptr[:Nmeas_obs]])
            cam_of_row = (first - i_i)//Nper
        Jc = Jc * (cam_of_row == icam)[:,None]
    Jp = Jd.copy(); Jp[:, ~keep] = 0
    return -np.linalg.lstsq(Jc, Jp, rcond=None)[0]


def board_problem(api, Ncameras=2, Nframes=6, lensmodel="LENSMODEL_OPENCV4", seed=41, W=6, H=5):
    from mrcal_amd.synthetic import make_calibration_problem
    oi, _ = make_calibration_problem(api, Ncameras=Ncameras, Nframes=Nframes, lensmodel=lensmodel,
                                     object_width_n=W, object_height_n=H, seed=seed)
    oi["observations_board"][1,2,1:3,2] = -1.      # outliers on input: all-zero rows
    return oi


def points_problem(api, seed=5):
    """discrete points only (no boards, no frames): 3 cameras (one at the reference), 12 points of which 2 fixed"""
    rng = np.random.RandomState(seed)
    Ncam, Np = 3, 12
    W, H = 4000, 2200
    intr = np.tile(np.array((1500., 1500., (W-1)/2., (H-1)/2., -0.01, 0.02, 1e-3, -2e-3)), (Ncam,1))
    rt_cam_ref = np.array(((0.01, -0.02, 0.03, -0.5, 0.02, 0.01), (-0.02, 0.01, 0.02, -1.0, -0.03, 0.02)))
    pts = np.column_stack((rng.uniform(-1, 2, Np), rng.uniform(-1, 1, Np), rng.uniform(4, 9, Np)))
    idx, obs = [], []
    for ip in range(Np):
        for ic in range(Ncam):
            idx.append((ip, ic, ic-1))
            obs.append((rng.uniform(800, 3000), rng.uniform(500, 1700), rng.uniform(0.5, 1.0)))
    obs = np.array(obs); obs[4,2] = -1.
    return dict(intrinsics=intr, lensmodel="LENSMODEL_OPENCV4",
                imagersizes=np.tile(np.array((W,H), dtype=np.int32), (Ncam,1)),
                rt_cam_ref=rt_cam_ref, points=pts, Npoints_fixed=2,
                observations_point=obs, indices_point_camintrinsics_camextrinsics=np.array(idx, dtype=np.int32),
                do_optimize_intrinsics_core=True, do_optimize_intrinsics_distortions=True,
                do_optimize_extrinsics=True, do_optimize_frames=True, do_optimize_calobject_warp=False,
                do_apply_regularization=True, do_apply_outlier_rejection=False, verbose=False)


CASES = (("boards", -1), ("boards", 0), ("boards", 1), ("boards-monocular", -1), ("boards-monocular", 0),
         ("boards-no-warp", -1), ("boards-no-warp", 1), ("points", -1), ("points", 0), ("points", 2))


def make_case(api, what):
    if what == "points":
        return points_problem(api)
    oi = board_problem(api, Ncameras=(1 if what == "boards-monocular" else 3))
    if what == "boards-no-warp":
        oi["do_optimize_calobject_warp"] = False
    return oi


# ------------------------------------------------------------------ CPU ---
def test_lapack_stub_is_lapack():
    """oracle/lapack_packed_stub.c (dpptrf_/dpptrs_, uplo='L') against numpy on SPD 6x6 matrices, through the
    packed layout the reference hands over (uncertainty.c:1501-1519)"""
    import os
    if not os.path.exists(REFLIB_PATH):
        pytest.skip("oracle/_ref/libmrcal_ref.so is not built")
    L = C.CDLL(REFLIB_PATH)
    rng = np.random.RandomState(3)
    for n in (1, 3, 6):
        A = rng.normal(size=(n+3, n)); A = A.T @ A
        ap = np.array([A[i,j] for j in range(n) for i in range(j, n)])     # column-major packed lower
        info = C.c_int(99)
        L.dpptrf_(C.c_char_p(b"L"), C.byref(C.c_int(n)), ap.ctypes.data_as(C.c_void_p), C.byref(info))
        assert info.value == 0
        Lc = np.linalg.cholesky(A)
        assert np.abs(ap - np.array([Lc[i,j] for j in range(n) for i in range(j, n)])).max() < 1e-13*np.abs(Lc).max()
        rhs = rng.normal(size=(2, n))
        b = rhs.copy()
        L.dpptrs_(C.c_char_p(b"L"), C.byref(C.c_int(n)), C.byref(C.c_int(2)), ap.ctypes.data_as(C.c_void_p),
                  b.ctypes.data_as(C.c_void_p), C.byref(C.c_int(n)), C.byref(info))
        assert info.value == 0
        assert np.abs(b - np.linalg.solve(A, rhs.T).T).max() < 1e-12*np.abs(b).max()
    # not positive definite: info = the order of the offending minor
    ap = np.array((1., 2., 1.))       # [[1,2],[2,1]]
    info = C.c_int(0)
    L.dpptrf_(C.c_char_p(b"L"), C.byref(C.c_int(2)), ap.ctypes.data_as(C.c_void_p), C.byref(info))
    assert info.value == 2


@pytest.mark.parametrize("what,icam", CASES)
def test_checker_matches_dense_expression(ref_api, what, icam):
    """the reference's function (compiled) == the dense lstsq, like the reference's own test asserts"""
    oi = make_case(ref_api, what)
    K  = ref_api.drt_cross_reprojection__dbpacked(icam_intrinsics=icam, **oi)
    Kd = dense_K(ref_api, oi, icam)
    assert K.shape == Kd.shape == (6, ref_api.num_states(**oi))
    assert np.abs(K).max() > 0
    assert np.abs(K - Kd).max() < 1e-9*np.abs(Kd).max()


def test_checker_refuses_what_the_reference_refuses(ref_api):
    oi = board_problem(ref_api)
    oi.update(do_optimize_extrinsics=False, do_optimize_frames=False, do_optimize_calobject_warp=False)
    with pytest.raises(RuntimeError):
        ref_api.drt_cross_reprojection__dbpacked(**oi)          # nothing to attribute a transform to
    oi = board_problem(ref_api)
    with pytest.raises(RuntimeError):
        ref_api.drt_cross_reprojection__dbpacked(icam_intrinsics=7, **oi)


# ------------------------------------------------------------------ GPU ---
@pytest.mark.gpu
@pytest.mark.parametrize("what,icam", CASES)
def test_matches_reference(amd, ref_api, what, icam):
    oi = make_case(amd._api, what)
    Ka = amd.drt_cross_reprojection__dbpacked(icam_intrinsics=icam, **oi)
    Kr = ref_api.drt_cross_reprojection__dbpacked(icam_intrinsics=icam, **oi)
    assert Ka.shape == Kr.shape
    # the same blocks are filled: nothing in the intrinsics columns, and extrinsics columns only with ccp
    Ni, Ne = amd.num_states_intrinsics(**oi), amd.num_states_extrinsics(**oi)
    assert not Ka[:, :Ni].any() and not Kr[:, :Ni].any()
    if icam < 0:
        assert not Ka[:, Ni:Ni+Ne].any() and not Kr[:, Ni:Ni+Ne].any()
    assert np.array_equal(np.abs(Ka).max(axis=0) > 0, np.abs(Kr).max(axis=0) > 0)     # (by columns)
    assert np.abs(Ka - Kr).max() < 1e-9*np.abs(Kr).max()
    Kd = dense_K(amd._api, oi, icam)
    assert np.abs(Ka - Kd).max() < 1e-9*np.abs(Kd).max()
    # the resident tier: J never leaves the device
    from mrcal_amd.resident import Problem
    with Problem(**copy_inputs(oi)) as p:
        Kp = p.drt_cross_reprojection__dbpacked(icam)
    assert np.abs(Kp - Ka).max() < 1e-12*np.abs(Ka).max()


@pytest.mark.gpu
def test_matches_reference_at_baseline_size(amd, ref_api):
    """configuration 1 (4 cameras x 400 frames OPENCV8, 7.2 M nonzeros), at the solved state: rrp and every ccp"""
    from mrcal_amd.synthetic import make_calibration_problem
    from mrcal_amd.resident import Problem
    oi, _ = make_calibration_problem(amd._api, Ncameras=4, Nframes=400, lensmodel="LENSMODEL_OPENCV8",
                                     object_width_n=10, object_height_n=10, seed=2)
    amd.optimize(**oi)
    with Problem(**copy_inputs(oi)) as p:
        for icam in (-1, 0, 3):
            Kp = p.drt_cross_reprojection__dbpacked(icam)
            Kr = ref_api.drt_cross_reprojection__dbpacked(icam_intrinsics=icam, **oi)
            assert np.abs(Kp - Kr).max() < 1e-9*np.abs(Kr).max(), icam
            # twice the same bits (fixed summation order)
            assert np.array_equal(Kp, p.drt_cross_reprojection__dbpacked(icam))


@pytest.mark.gpu
def test_error_behaviour(amd):
    oi = board_problem(amd._api)
    oi.update(do_optimize_extrinsics=False, do_optimize_frames=False, do_optimize_calobject_warp=False)
    with pytest.raises(RuntimeError):
        amd.drt_cross_reprojection__dbpacked(**oi)
    oi = board_problem(amd._api)
    with pytest.raises(RuntimeError):
        amd.drt_cross_reprojection__dbpacked(icam_intrinsics=7, **oi)
    # a malformed Jt is refused before any kernel walks it (the C entry point takes the caller's cholmod_sparse)
    def decreasing(J):  J.indptr[5] = J.indptr[4] - 1
    def out_of_range(J): J.indices[7] = J.shape[1]
    def negative(J):    J.indices[3] = -1
    for tamper in (decreasing, out_of_range, negative):
        with pytest.raises(RuntimeError, match="malformed Jt"):
            amd.drt_cross_reprojection__dbpacked(_tamper_with_J=tamper, **oi)
    assert np.isfinite(amd.drt_cross_reprojection__dbpacked(**oi)).all()
    # the intrinsics locked: ccp cannot tell the cameras apart (uncertainty.c:1190-1195), rrp works
    oi.update(do_optimize_intrinsics_core=False, do_optimize_intrinsics_distortions=False)
    with pytest.raises(RuntimeError):
        amd.drt_cross_reprojection__dbpacked(icam_intrinsics=0, **oi)
    assert np.isfinite(amd.drt_cross_reprojection__dbpacked(**oi)).all()
