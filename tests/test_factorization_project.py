"""The CHOLMOD_factorization equivalent and the stand-alone projection, on the GPU.

  - the reference's own known-answer test of the factorization
    (test/test-CHOLMOD-factorization.py: a 4x3 J, solve == dense solve, 1e-6)
  - the factorization optimizer_callback() returns, on calibration problems,
    against a dense numpy solve of JtJ
  - mrcal_project() for every lens model against the reference's"""
import ctypes as C
import numpy as np
import pytest
from scipy.sparse import csr_matrix

from conftest import relative_error
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd._cabi import Lensmodel

pytestmark = pytest.mark.gpu


def test_reference_known_answer(amd):
    # test/test-CHOLMOD-factorization.py:20-52
    indptr  = np.array([0, 2, 3, 6, 8])
    indices = np.array([0, 2, 2, 0, 1, 2, 1, 2])
    data    = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=float)
    J  = csr_matrix((data, indices, indptr))
    bt = np.array(((1., 5., 3.), (2., -2., -8)))
    F  = amd.CHOLMOD_factorization(J)
    xt = F.solve_xt_JtJ_bt(bt)
    Jd = J.toarray()
    xt_ref = np.linalg.solve(Jd.T @ Jd, bt.T).T
    assert relative_error(xt, xt_ref).max() < 1e-6
    assert 0. < F.rcond() <= 1.
    # 1-dimensional and 3-dimensional bt
    assert relative_error(F.solve_xt_JtJ_bt(bt[0]), xt_ref[0]).max() < 1e-6
    bt3 = np.ascontiguousarray(np.stack((bt, 2*bt)))
    assert relative_error(F.solve_xt_JtJ_bt(bt3), np.stack((xt_ref, 2*xt_ref))).max() < 1e-6
    with pytest.raises(NotImplementedError):
        F.solve_xt_JtJ_bt(bt, sys="P")
    with pytest.raises(RuntimeError):
        F.solve_xt_JtJ_bt(bt.astype(np.float32))


def test_singular_is_an_error_and_None_from_callback(amd):
    J = csr_matrix(np.array(((1., 2., 0.), (2., 4., 0.), (0., 0., 0.))))
    with pytest.raises(RuntimeError):
        amd.CHOLMOD_factorization(J)


@pytest.mark.parametrize("lensmodel,Ncam,Nf,with_points", (("LENSMODEL_OPENCV4", 2, 5, False),
                                                            ("LENSMODEL_OPENCV8", 3, 6, True),
                                                            ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120", 1, 8, False)))
def test_callback_factorization_solves(amd, lensmodel, Ncam, Nf, with_points):
    oi, _ = make_calibration_problem(amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel=lensmodel,
                                     object_width_n=8, object_height_n=7, seed=21)
    if with_points:
        from test_callback_parity import _with_points
        oi = _with_points(oi, np.random.RandomState(2))
        # (_with_points() leaves two points with a single live observation: their range
        #  is free and JtJ singular, cond 7e20 - numpy's Cholesky fails on it too.
        #  Here every point is seen twice)
        oi["observations_point"][:,2] = np.abs(oi["observations_point"][:,2]) + 0.5
    if "SPLINED" in lensmodel:
        oi["do_optimize_intrinsics_core"] = False
    b, x, J, F = amd.optimizer_callback(**oi)
    assert F is not None
    Jd = J.toarray()
    N  = Jd.T @ Jd
    rng = np.random.RandomState(4)
    # right-hand sides with a known solution. JtJ of a calibration problem is
    # badly conditioned (1e12 and beyond): the solution is compared where the
    # conditioning allows it, the residual always (scaled like a backward error)
    xtrue = rng.normal(size=(3, J.shape[1]))
    bt = np.ascontiguousarray(xtrue @ N)
    xt = F.solve_xt_JtJ_bt(bt)
    resid = np.abs(xt @ N - bt).max() / (np.abs(N).max()*np.abs(xt).max())
    assert resid < 1e-10
    cond = np.linalg.cond(N)
    if cond < 1e10:
        assert np.abs(xt - xtrue).max() < 1e-14*cond*10*np.abs(xtrue).max()
    assert 0. < F.rcond() <= 1.


MODELS = [
    ("LENSMODEL_PINHOLE",       (1512., 1112, 500., 333.)),
    ("LENSMODEL_STEREOGRAPHIC", (1512., 1112, 500., 333.)),
    ("LENSMODEL_LONLAT",        (1200., 1150, 500., 333.)),
    ("LENSMODEL_LATLON",        (1200., 1150, 500., 333.)),
    ("LENSMODEL_OPENCV4",  (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002)),
    ("LENSMODEL_OPENCV8",  (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019, 0.014, -0.056, 0.050)),
    ("LENSMODEL_OPENCV12", (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019, 0.014, -0.056, 0.050,
                            0.003, -0.002, 0.001, 0.004)),
    ("LENSMODEL_CAHVOR",   (4842.918, 4842.771, 1970.528, 1085.302, -0.001, 0.002, -0.637, -0.002, 0.016)),
    ("LENSMODEL_CAHVORE_linearity=0.40", (4842.918, 4842.771, 1970.528, 1085.302, -0.001, 0.002, -0.637, -0.002, 0.016, 1e-2, 2e-2, 3e-2)),
]


def _ref_project(ref_api, lensmodel, intr, p):
    clib = ref_api.clib
    clib.mrcal_project.restype  = C.c_bool
    clib.mrcal_project.argtypes = [C.c_void_p]*4 + [C.c_int, C.POINTER(Lensmodel), C.c_void_p]
    m = Lensmodel()
    assert clib.mrcal_lensmodel_from_name(C.byref(m), lensmodel.encode())
    N, Ni = p.shape[0], len(intr)
    q, g, gi = np.zeros((N,2)), np.zeros((N,2,3)), np.zeros((N,2,Ni))
    assert clib.mrcal_project(q.ctypes.data, g.ctypes.data, gi.ctypes.data, p.ctypes.data, N, C.byref(m), intr.ctypes.data)
    return q, g, gi


@pytest.mark.parametrize("lensmodel,intrinsics", MODELS, ids=[m[0] for m in MODELS])
def test_project_matches_reference(amd, ref_api, lensmodel, intrinsics):
    rng = np.random.RandomState(3)
    N = 1000
    p = np.ascontiguousarray(np.column_stack((rng.uniform(-1.2, 1.2, N), rng.uniform(-1.0, 1.0, N), rng.uniform(0.8, 6.0, N))))
    intr = np.array(intrinsics, dtype=float)
    q, g, gi = amd.project(p, lensmodel, intr, get_gradients=True)
    qr, gr, gir = _ref_project(ref_api, lensmodel, intr, p)
    assert relative_error(q, qr).max() < 1e-6
    assert relative_error(g, gr).max() < 1e-6
    assert relative_error(gi, gir).max() < 1e-6
    assert np.array_equal(amd.project(p, lensmodel, intr), q)
    # leading dimensions are kept
    assert amd.project(p.reshape(10,100,3), lensmodel, intr).shape == (10,100,2)


def test_project_splined_matches_reference(amd, ref_api):
    lensmodel = "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=200"
    rng = np.random.RandomState(5)
    intr = np.concatenate(((1500., 1800., 1499.5, 999.5), rng.uniform(-0.05, 0.05, 2*11*8)))
    N = 500
    p = np.ascontiguousarray(np.column_stack((rng.uniform(-3, 3, N), rng.uniform(-2, 2, N), rng.uniform(0.3, 3.0, N))))
    q, g, gi = amd.project(p, lensmodel, intr, get_gradients=True)
    qr, gr, gir = _ref_project(ref_api, lensmodel, intr, p)
    assert relative_error(q, qr).max() < 1e-6
    assert relative_error(g, gr).max() < 1e-6
    assert relative_error(gi, gir).max() < 1e-6
