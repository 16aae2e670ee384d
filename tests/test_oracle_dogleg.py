"""Pins the CHECKER's restated solver (oracle/dogleg_restated.c: libdogleg +
CHOLMOD are third-party, not vendored by the reference and not installed) to
independent numerics. CPU only.

  - dogleg_restated_solve_JtJ (the sparse up-looking Cholesky) against
    numpy.linalg.solve on the dense JtJ: the reference's own known-answer case
    (test/test-CHOLMOD-factorization.py:20-52), random sparse matrices with the
    arrowhead structure of calibration problems, and the Jacobian of the
    reference's own callback on a calibration problem
  - a matrix that is not positive definite is reported as such
  - dogleg_optimize2 driven by a Python dogleg_callback_t on small nonlinear
    least-squares problems (Rosenbrock, Powell's singular function, an
    exponential fit) lands on the optimum scipy.optimize.least_squares finds
  - the reference's mrcal_optimize() through the restated solver reaches the
    stationary point scipy finds for the same cost function on the reference's
    callback (a small calibration)
  - the default parameters are libdogleg's (the ones mrcal.c:6296-6299 overrides on top of)
"""
import ctypes as C
import os
import numpy as np
import pytest
from scipy.sparse import csr_matrix, random as sparse_random

from conftest import ROOT, relative_error

LIB = os.path.join(ROOT, "oracle", "_build", "liboracle_dogleg.so")


class CholmodSparse(C.Structure):
    # oracle/stubs/dogleg.h: field order of SuiteSparse's cholmod_sparse
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t),
                ("p", C.c_void_p), ("i", C.c_void_p), ("nz", C.c_void_p),
                ("x", C.c_void_p), ("z", C.c_void_p),
                ("stype", C.c_int), ("itype", C.c_int), ("xtype", C.c_int),
                ("dtype", C.c_int), ("sorted", C.c_int), ("packed", C.c_int)]


class Parameters2(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("dogleg_debug", C.c_int),
                ("trustregion0", C.c_double),
                ("trustregion_decrease_factor", C.c_double), ("trustregion_decrease_threshold", C.c_double),
                ("trustregion_increase_factor", C.c_double), ("trustregion_increase_threshold", C.c_double),
                ("Jt_x_threshold", C.c_double), ("update_threshold", C.c_double),
                ("trustregion_threshold", C.c_double)]


CALLBACK = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(CholmodSparse), C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.skip("oracle/_build/liboracle_dogleg.so is not built (make -C oracle)")
    L = C.CDLL(LIB)
    L.dogleg_restated_solve_JtJ.restype  = C.c_bool
    L.dogleg_restated_solve_JtJ.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dogleg_optimize2.restype  = C.c_double
    L.dogleg_optimize2.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, CALLBACK, C.c_void_p,
                                   C.POINTER(Parameters2), C.c_void_p]
    L.dogleg_getDefaultParameters.argtypes = [C.POINTER(Parameters2)]
    return L


def solve_JtJ(lib, J, bt):
    """bt: (Nrhs,Nstate). Solves in place through the restated Cholesky; J is
    scipy CSR (rows = measurements), which IS the compressed-column Jt the
    solver takes (mrcal.c:4461-4520)"""
    J = csr_matrix(J)
    J.sort_indices()
    p = np.ascontiguousarray(J.indptr,  dtype=np.int32)
    i = np.ascontiguousarray(J.indices, dtype=np.int32)
    x = np.ascontiguousarray(J.data,    dtype=np.float64)
    b = np.ascontiguousarray(np.atleast_2d(bt), dtype=np.float64).copy()
    ok = lib.dogleg_restated_solve_JtJ(b.ctypes.data, b.shape[0], J.shape[1], J.shape[0],
                                       p.ctypes.data, i.ctypes.data, x.ctypes.data)
    return ok, b


def test_reference_known_answer(lib):
    # test/test-CHOLMOD-factorization.py:20-52
    indptr  = np.array([0, 2, 3, 6, 8])
    indices = np.array([0, 2, 2, 0, 1, 2, 1, 2])
    data    = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=float)
    J  = csr_matrix((data, indices, indptr))
    bt = np.array(((1., 5., 3.), (2., -2., -8)))
    ok, xt = solve_JtJ(lib, J, bt)
    assert ok
    Jd = J.toarray()
    xt_ref = np.linalg.solve(Jd.T @ Jd, bt.T).T
    assert relative_error(xt, xt_ref).max() < 1e-6   # the reference test's bar
    assert np.abs(xt - xt_ref).max() < 1e-12*np.abs(xt_ref).max()


@pytest.mark.parametrize("seed", range(4))
def test_random_sparse_against_dense_numpy(lib, seed):
    rng = np.random.RandomState(seed)
    Nstate, Nmeas = 60 + 7*seed, 400
    J = sparse_random(Nmeas, Nstate, density=0.08, random_state=rng, format="csr")
    J = J + csr_matrix((np.ones(Nstate), (np.arange(Nstate), np.arange(Nstate))), shape=(Nmeas, Nstate))
    bt = rng.normal(size=(3, Nstate))
    ok, xt = solve_JtJ(lib, J, bt)
    assert ok
    Jd = J.toarray()
    xt_ref = np.linalg.solve(Jd.T @ Jd, bt.T).T
    assert np.abs(xt - xt_ref).max() < 1e-9*np.abs(xt_ref).max()


def test_arrowhead_against_dense_numpy(lib):
    """the structure of a calibration problem: each row touches a dense
    'camera' block and one of many small 'frame' blocks"""
    rng = np.random.RandomState(10)
    Nc, Nf = 20, 40
    Nstate = Nc + 6*Nf
    rows = []
    for f in range(Nf):
        for _ in range(12):
            r = np.zeros(Nstate)
            r[rng.choice(Nc, 5, replace=False)] = rng.normal(size=5)
            r[Nc+6*f:Nc+6*f+6] = rng.normal(size=6)
            rows.append(r)
    Jd = np.array(rows)
    bt = rng.normal(size=(2, Nstate))
    ok, xt = solve_JtJ(lib, csr_matrix(Jd), bt)
    assert ok
    xt_ref = np.linalg.solve(Jd.T @ Jd, bt.T).T
    assert np.abs(xt - xt_ref).max() < 1e-9*np.abs(xt_ref).max()


def test_not_positive_definite_is_reported(lib):
    Jd = np.array(((1., 2., 0.), (2., 4., 0.), (0., 0., 0.)))
    ok, _ = solve_JtJ(lib, csr_matrix(Jd), np.ones((1,3)))
    assert not ok


def test_calibration_jacobian_against_dense_numpy(lib, ref_api):
    """JtJ of the reference's own callback on a calibration problem (with
    regularization: otherwise unobserved variables make it singular)"""
    from mrcal_amd.synthetic import make_calibration_problem
    oi, _ = make_calibration_problem(ref_api, Ncameras=2, Nframes=8, object_width_n=5, object_height_n=4, seed=3)
    _, x, J, _ = ref_api.optimizer_callback(no_factorization=True, **oi)
    g = J.T @ x
    ok, d = solve_JtJ(lib, J, g[None,:])
    assert ok
    Jd = J.toarray()
    d_ref = np.linalg.solve(Jd.T @ Jd, g)
    assert np.abs(d[0] - d_ref).max() < 1e-7*np.abs(d_ref).max()
    # and against a QR least-squares solve, which never forms JtJ
    d_qr = np.linalg.lstsq(Jd, x, rcond=None)[0]
    assert np.abs(d[0] - d_qr).max() < 1e-5*np.abs(d_qr).max()


# --------------------------------------------------------------------------
# the dog-leg loop on small problems with known optima

def _run_dogleg(lib, residual, jacobian, p0, **params):
    """dogleg_optimize2 with a Python callback. jacobian(p) is dense (Nmeas,
    Nstate); handed over as a FULL compressed-column Jt"""
    p = np.array(p0, dtype=float)
    Nstate = len(p)
    Nmeas  = len(residual(p))
    ncalls = [0]

    def cb(pp, xx, Jt, cookie):
        ncalls[0] += 1
        pv = np.ctypeslib.as_array(pp, shape=(Nstate,))
        xv = np.ctypeslib.as_array(xx, shape=(Nmeas,))
        xv[:] = residual(pv)
        if Jt:
            jt = Jt.contents
            P = np.ctypeslib.as_array(C.cast(jt.p, C.POINTER(C.c_int32)), shape=(Nmeas+1,))
            I = np.ctypeslib.as_array(C.cast(jt.i, C.POINTER(C.c_int32)), shape=(Nmeas*Nstate,))
            X = np.ctypeslib.as_array(C.cast(jt.x, C.POINTER(C.c_double)), shape=(Nmeas*Nstate,))
            Jd = jacobian(pv)
            P[:] = np.arange(Nmeas+1)*Nstate
            I[:] = np.tile(np.arange(Nstate), Nmeas)
            X[:] = Jd.ravel()

    par = Parameters2()
    lib.dogleg_getDefaultParameters(C.byref(par))
    for k, v in params.items():
        setattr(par, k, v)
    cbc = CALLBACK(cb)
    norm2 = lib.dogleg_optimize2(p.ctypes.data, Nstate, Nmeas, Nmeas*Nstate, cbc, None, C.byref(par), None)
    return p, norm2, ncalls[0]


def test_default_parameters_are_libdoglegs(lib):
    par = Parameters2()
    lib.dogleg_getDefaultParameters(C.byref(par))
    # libdogleg's published defaults (dogleg.c: DOGLEG_DEFAULT_*), the values
    # mrcal.c:6296-6299 then overrides
    assert par.max_iterations == 100
    assert par.trustregion0 == 1e3
    assert (par.trustregion_decrease_factor, par.trustregion_decrease_threshold) == (0.1, 0.25)
    assert (par.trustregion_increase_factor, par.trustregion_increase_threshold) == (2.0, 0.75)
    assert par.Jt_x_threshold == 1e-8 and par.update_threshold == 1e-8 and par.trustregion_threshold == 1e-8


PROBLEMS = {
    "rosenbrock": (lambda p: np.array((10.*(p[1]-p[0]**2), 1.-p[0])),
                   lambda p: np.array(((-20.*p[0], 10.), (-1., 0.))),
                   (-1.2, 1.0)),
    "powell_singular": (lambda p: np.array((p[0]+10*p[1], np.sqrt(5.)*(p[2]-p[3]), (p[1]-2*p[2])**2, np.sqrt(10.)*(p[0]-p[3])**2)),
                        lambda p: np.array(((1., 10., 0., 0.),
                                            (0., 0., np.sqrt(5.), -np.sqrt(5.)),
                                            (0., 2*(p[1]-2*p[2]), -4*(p[1]-2*p[2]), 0.),
                                            (2*np.sqrt(10.)*(p[0]-p[3]), 0., 0., -2*np.sqrt(10.)*(p[0]-p[3])))),
                        (3., -1., 0., 1.)),
}
_t = np.linspace(0., 4., 30)
_y = 2.5*np.exp(-1.3*_t) + 0.5 + 0.01*np.cos(37.*_t)
PROBLEMS["exponential_fit"] = (lambda p: p[0]*np.exp(p[1]*_t) + p[2] - _y,
                               lambda p: np.stack((np.exp(p[1]*_t), p[0]*_t*np.exp(p[1]*_t), np.ones_like(_t)), axis=-1),
                               (1., -0.5, 0.))


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_dogleg_loop_finds_the_optimum_scipy_finds(lib, name):
    from scipy.optimize import least_squares
    residual, jacobian, p0 = PROBLEMS[name]
    p, norm2, ncalls = _run_dogleg(lib, residual, jacobian, p0, max_iterations=1000)
    ref = least_squares(residual, p0, jac=jacobian, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert abs(norm2 - float(residual(p) @ residual(p))) <= 1e-12*max(norm2, 1e-300)
    if name == "powell_singular":
        # singular Jacobian at the optimum (0): slow, cost -> 0
        assert norm2 < 1e-12 and np.abs(p).max() < 1e-3
    else:
        assert np.abs(p - ref.x).max() < 1e-6*max(1., np.abs(ref.x).max())
        assert abs(norm2 - 2*ref.cost) < 1e-9*max(1., 2*ref.cost)
    assert ncalls < 1000


def test_reference_optimize_through_restated_solver_vs_scipy(ref_api):
    """the reference's mrcal_optimize() + restated dog-leg ends at a point that
    scipy's trust-region least squares, on the reference's callback, accepts as
    the optimum"""
    from scipy.optimize import least_squares
    from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
    oi, _ = make_calibration_problem(ref_api, Ncameras=2, Nframes=10, object_width_n=5, object_height_n=5,
                                     seed=5, make_outliers=False)
    oi["do_apply_outlier_rejection"] = False
    oa = copy_inputs(oi)
    s = ref_api.optimize(**oa)

    ob = copy_inputs(oi)
    b0 = ref_api.optimizer_callback(no_jacobian=True, no_factorization=True, **ob)[0]

    def at(b):
        o = copy_inputs(oi)
        bb = b.copy()
        ref_api.unpack_state(bb, **o)
        n = 0
        Ni = o["intrinsics"].size
        o["intrinsics"][...]   = bb[n:n+Ni].reshape(o["intrinsics"].shape); n += Ni
        o["rt_cam_ref"][...]   = bb[n:n+o["rt_cam_ref"].size].reshape(o["rt_cam_ref"].shape); n += o["rt_cam_ref"].size
        o["rt_ref_frame"][...] = bb[n:n+o["rt_ref_frame"].size].reshape(o["rt_ref_frame"].shape); n += o["rt_ref_frame"].size
        o["calobject_warp"][...] = bb[n:n+2]
        return o

    def fun(b): return ref_api.optimizer_callback(no_jacobian=True, no_factorization=True, **at(b))[1]
    def jac(b): return ref_api.optimizer_callback(no_factorization=True, **at(b))[2]
    # scipy started at the restated solver's answer cannot improve on it, and
    # the gradient there is zero to the solver's thresholds
    b1 = s["b_packed"]
    x1 = fun(b1); J1 = jac(b1)
    Nmeas = len(x1)
    assert abs(np.sqrt(x1 @ x1 / Nmeas) - s["rms_reproj_error__pixels"]) < 1e-12
    g = J1.T @ x1
    assert np.abs(g).max() < 1e-5*np.sqrt((J1.data**2).sum())*np.linalg.norm(x1)
    r = least_squares(fun, b1, jac=jac, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14, x_scale=1.0, max_nfev=30)
    assert x1 @ x1 - 2*r.cost < 1e-9*(x1 @ x1)
    assert np.abs(r.x - b1).max() < 1e-4
    # and scipy started at the seed does not find anything better
    r0 = least_squares(fun, b0, jac=jac, method="trf", x_scale=1.0, max_nfev=60)
    assert x1 @ x1 <= 2*r0.cost*(1 + 1e-9)


def test_restated_factorization_follows_a_changing_pattern(ref_api, tmp_path):
    """The splined lens models move a row's columns with the corner (mrcal.c:4718-4817), so the pattern of Jt
    changes between evaluations. CHOLMOD's simplicial factorization (what libdogleg runs: supernodal = 0) is
    correct for any pattern; the restatement redoes its static symbolic analysis when the pattern moved. Until
    round 4 it did not, found "not positive definite" matrices that numpy.linalg.cholesky factors without trouble
    (52 times on this problem: tools/diag_splined_pd.py, profiles/r04_splined_checker_defect.txt), and returned
    1.3026 px with 11 outliers where the stationary point is at 1.1814 px with 15. Sweep seed 23, case 22 of
    tools/fuzz_parity.py, built by the reference's own library: nothing here needs a GPU"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity, arbiter
    from mrcal_amd.synthetic import copy_inputs
    rng = np.random.RandomState(23)
    for ic, what, oi, *_ in fuzz_parity.board_cases(23, rng, ref_api):
        if ic == 22: break
    assert "SPLINED" in what
    # the pattern does move on this problem: the seed's Jacobian and the solution's
    o = copy_inputs(oi)
    J0 = ref_api.optimizer_callback(no_factorization=True, **copy_inputs(o))[2]
    os.environ["DOGLEG_RESTATED_DUMP_NOTPD"] = str(tmp_path / "notpd_")
    try:
        s = ref_api.optimize(**o)
    finally:
        del os.environ["DOGLEG_RESTATED_DUMP_NOTPD"]
    J1 = ref_api.optimizer_callback(no_factorization=True, **copy_inputs(o))[2]
    assert np.array_equal(J0.indptr, J1.indptr) and not np.array_equal(J0.indices, J1.indices)
    assert list(tmp_path.glob("notpd_*.bin")) == []         # JtJ was never declared "not positive definite"
    assert abs(s["rms_reproj_error__pixels"] - 1.18136) < 1e-4 and s["Noutliers_board"] == 15
    st, cost, _ = arbiter.stationarity(ref_api, o)
    assert st < 1e-7, st
    assert arbiter.least_squares_gain(ref_api, o) < 1e-9
