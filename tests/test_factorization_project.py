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
    with pytest.raises(RuntimeError, match="Unknown sys"):
        F.solve_xt_JtJ_bt(bt, sys="Q")
    with pytest.raises(RuntimeError):
        F.solve_xt_JtJ_bt(bt.astype(np.float32))
    # degenerate input: returned as it is (mrcal-pywrap.c:524-530)
    assert F.solve_xt_JtJ_bt(np.zeros((0,3))).shape == (0,3)


def test_singular_is_an_error_and_None_from_callback(amd):
    J = csr_matrix(np.array(((1., 2., 0.), (2., 4., 0.), (0., 0., 0.))))
    with pytest.raises(RuntimeError):
        amd.CHOLMOD_factorization(J)


def test_wrong_structure_is_an_error(amd):
    """the partition that optimizer_callback() declares must be the matrix's: a row that touches two frame blocks, or a
    column beyond the state, is reported (by the assembly kernel itself), not factored"""
    import scipy.sparse
    from mrcal_amd._factorization import CHOLMOD_factorization
    rng = np.random.default_rng(0)
    Nc, Nfb = 5, 4                                  # 5 shared variables, 4 frame blocks of 6
    Nstate = Nc + 6*Nfb
    rows = []
    for f in range(Nfb):
        for _ in range(40):
            r = np.zeros(Nstate); r[:Nc] = rng.normal(size=Nc); r[Nc + 6*f: Nc + 6*f + 6] = rng.normal(size=6)
            rows.append(r)
    J = scipy.sparse.csr_matrix(np.array(rows))
    f = CHOLMOD_factorization(J, _partition=(Nc, Nfb, 0, 0))        # fine
    assert f.solve_xt_JtJ_bt(np.ones((1, Nstate))).shape == (1, Nstate)
    bad = np.array(rows); bad[7, Nc + 6*2] = 1.0                     # row 7 (frame 0) now touches frame 2 as well
    with pytest.raises(RuntimeError, match="structure"):
        CHOLMOD_factorization(scipy.sparse.csr_matrix(bad), _partition=(Nc, Nfb, 0, 0))
    Jb = scipy.sparse.csr_matrix(np.array(rows)); Jb.indices = Jb.indices.copy(); Jb.indices[3] = Nstate + 5
    with pytest.raises(RuntimeError, match="structure"):
        CHOLMOD_factorization(Jb, _partition=(Nc, Nfb, 0, 0))


@pytest.mark.parametrize("lensmodel,Ncam,Nf,with_points", (("LENSMODEL_OPENCV4", 2, 5, False),
                                                            ("LENSMODEL_OPENCV8", 3, 6, True),
                                                            ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120", 1, 8, False),
                                                            # a camera block past what the LDS triangular solve holds
                                                            # (234, and 2*160 + 6 = 326: the blocked one)
                                                            ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=13_Ny=9_fov_x_deg=120", 1, 8, False),
                                                            ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=10_Ny=8_fov_x_deg=120", 2, 8, False)))
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
    # rcond() IS cholmod_rcond()'s definition for an LL' factor (mrcal-pywrap.c:576-592 -> cholmod_rcond():
    # (min L_ii / max L_ii)^2), evaluated on THIS factor: numpy's Cholesky of the matrix in this factor's order
    # (sys='P' applied to 0..Nstate-1 is the order) gives the same number. CHOLMOD's differs from it only through
    # its own (AMD) ordering
    order = F.solve_xt_JtJ_bt(np.arange(J.shape[1], dtype=float), sys="P").round().astype(int)
    assert sorted(order) == list(range(J.shape[1]))
    dL = np.diag(np.linalg.cholesky(N[np.ix_(order, order)]))
    expected = (dL.min()/dL.max())**2
    assert abs(F.rcond() - expected) < (1e-6 if cond < 1e10 else 0.5)*expected, (F.rcond(), expected, cond)


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


def _factor_dense(F, N):
    """L (dense, lower) and the permutation matrix P out of the sys= interface:
    column i of L^-1 is solve(e_i, 'L'); P from solve(I, 'P')"""
    I  = np.eye(N)
    Pm = F.solve_xt_JtJ_bt(I, sys="P").T           # rows of the result are P e_i: columns of P
    Linv = F.solve_xt_JtJ_bt(I, sys="L").T
    return np.linalg.inv(Linv), Pm


@pytest.mark.parametrize("lensmodel,Ncam,Nf,with_points", (("LENSMODEL_OPENCV4", 2, 5, False),
                                                            ("LENSMODEL_OPENCV8", 3, 10, True),
                                                            ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=13_Ny=9_fov_x_deg=120", 1, 8, False)))
def test_sys_variants(amd, ref_api, lensmodel, Ncam, Nf, with_points):
    """the sys= surface of CHOLMOD_factorization.solve_xt_JtJ_bt
    (mrcal-pywrap.c:467-493) on the factorization optimizer_callback() returns:
    L L^T == P JtJ P^T densely, every system against dense numpy, and the
    sequence mrcal's projection uncertainty runs (model_analysis.py:837-843)"""
    from test_callback_parity import _with_points
    rng = np.random.RandomState(3)
    oi, _ = make_calibration_problem(amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel=lensmodel,
                                     object_width_n=7, object_height_n=6, seed=8)
    if with_points: oi = _with_points(oi, rng)
    if "SPLINED" in lensmodel: oi["do_optimize_intrinsics_core"] = False
    _, x, J, F = amd.optimizer_callback(**oi)
    assert F is not None
    N = J.shape[1]
    JtJ = (J.T @ J).toarray()
    L, Pm = _factor_dense(F, N)
    assert np.allclose(np.triu(L, 1), 0, atol=1e-9*np.abs(L).max())                 # lower triangular
    assert np.array_equal(np.sort(Pm @ np.arange(N)), np.arange(N))                # a permutation
    assert np.abs(L @ L.T - Pm @ JtJ @ Pm.T).max() < 1e-9*np.abs(JtJ).max()
    B = rng.normal(size=(3, N))
    # every system through its residual (JtJ of a problem with a few discrete points is
    # badly conditioned: the solutions of two correct solvers differ, their residuals do not)
    ops = {"A": JtJ, "L": L, "LD": L, "Lt": L.T, "DLt": L.T, "LDLt": L @ L.T, "D": np.eye(N), "P": Pm.T, "Pt": Pm}
    for sys, M in ops.items():
        got = F.solve_xt_JtJ_bt(B, sys=sys)
        res = (M @ got.T).T - B
        assert np.abs(res).max() <= 1e-9*np.abs(M).max()*np.abs(got).max()/max(1.0, np.abs(B).max()) + 1e-9, sys
        assert np.abs(F.solve_xt_JtJ_bt(B, sys="CHOLMOD_" + sys) - got).max() == 0
    # the projection-uncertainty sequence
    A1 = F.solve_xt_JtJ_bt(B,  sys="P")
    A2 = F.solve_xt_JtJ_bt(A1, sys="L")
    A3 = F.solve_xt_JtJ_bt(A2, sys="D")
    Var = A2 @ A3.T
    Var_ref = B @ F.solve_xt_JtJ_bt(B).T
    assert np.abs(Var - Var_ref).max() < 1e-9*np.abs(Var_ref).max()


def test_solve_batch_is_row_independent(amd):
    """solve_xt_JtJ_bt takes any number of right-hand sides (mrcal's uncertainty grids pass thousands:
    model_analysis.py:837-843); here they are solved side by side on the device. A row's answer has the
    same bits alone, in a small batch and in a big one"""
    oi, _ = make_calibration_problem(amd._api, Ncameras=2, Nframes=12, lensmodel="LENSMODEL_OPENCV8",
                                     object_width_n=7, object_height_n=6, seed=5)
    _, x, J, F = amd.optimizer_callback(**oi)
    N = J.shape[1]
    B = np.random.RandomState(0).normal(size=(300, N))
    for sys in ("A", "L", "Lt", "P"):
        big = F.solve_xt_JtJ_bt(B, sys=sys)
        assert big.shape == B.shape
        for i in (0, 1, 150, 299):
            assert np.array_equal(F.solve_xt_JtJ_bt(B[i], sys=sys), big[i]), (sys, i)
        assert np.array_equal(F.solve_xt_JtJ_bt(B[10:17], sys=sys), big[10:17]), sys
    JtJ = (J.T @ J).toarray()
    Xs = F.solve_xt_JtJ_bt(B)
    assert np.abs(Xs @ JtJ - B).max() < 1e-9*np.abs(JtJ).max()*np.abs(Xs).max()


def test_Jt_x_and_A_Jt_J_At(amd):
    """mrcal._mrcal_npsp._Jt_x / _A_Jt_J_At / _A_Jt_J_At__2 (mrcal-genpywrap.py:477-731):
    from the p, i, x arrays like the reference's, and on the J resident with a factorization"""
    rng = np.random.RandomState(4)
    oi, _ = make_calibration_problem(amd._api, Ncameras=2, Nframes=6, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=5, object_height_n=4, seed=9)
    _, x, J, F = amd.optimizer_callback(**oi)
    Nmeas, Nstate = J.shape
    out = np.zeros((Nstate,))
    amd._Jt_x(J.indptr, J.indices, J.data, x, out=out)
    ref = J.T @ x
    assert np.abs(out - ref).max() < 1e-12*np.abs(ref).max()
    assert np.abs(F._Jt_x(x) - ref).max() < 1e-12*np.abs(ref).max()
    with pytest.raises(RuntimeError, match="must match the number of rows"):
        amd._Jt_x(J.indptr, J.indices, J.data, x[:-1], out=out)
    Nlead = amd.num_measurements_boards(**oi)
    Jl = J[:Nlead].toarray()
    for Nx in (2, 3):
        A = rng.normal(size=(Nx, Nstate))
        ref = A @ Jl.T @ Jl @ A.T
        got = amd._A_Jt_J_At(A, J.indptr, J.indices, J.data, Nleading_rows_J=Nlead)
        assert np.abs(got - ref).max() < 1e-12*np.abs(ref).max()
        assert np.abs(F._A_Jt_J_At(A, Nleading_rows_J=Nlead) - ref).max() < 1e-12*np.abs(ref).max()
    A = rng.normal(size=(2, Nstate))
    assert np.array_equal(amd._A_Jt_J_At__2(A, J.indptr, J.indices, J.data, Nleading_rows_J=Nlead),
                          amd._A_Jt_J_At(A, J.indptr, J.indices, J.data, Nleading_rows_J=Nlead))
    with pytest.raises(RuntimeError, match="Nleading_rows_J must be passed"):
        amd._A_Jt_J_At(A, J.indptr, J.indices, J.data)
    # the same bits every time: no floating-point atomics behind either product (round 4)
    for rep in range(3):
        out2 = np.zeros((Nstate,))
        amd._Jt_x(J.indptr, J.indices, J.data, x, out=out2)
        assert np.array_equal(out2, out) and np.array_equal(F._Jt_x(x), out)
        assert np.array_equal(amd._A_Jt_J_At(A, J.indptr, J.indices, J.data, Nleading_rows_J=Nlead),
                              amd._A_Jt_J_At__2(A, J.indptr, J.indices, J.data, Nleading_rows_J=Nlead))
    # a malformed CSR is an error, not an out-of-bounds write on the device
    bad = J.indices.copy(); bad[7] = Nstate + 3
    with pytest.raises(RuntimeError, match="column index"):
        amd._Jt_x(J.indptr, bad, J.data, x, out=out)
    with pytest.raises(RuntimeError, match="column index"):
        amd._A_Jt_J_At(A, J.indptr, bad, J.data, Nleading_rows_J=Nlead)
    badp = J.indptr.copy(); badp[5] = badp[6] + 1
    with pytest.raises(RuntimeError, match="rowptr"):
        amd._Jt_x(badp, J.indices, J.data, x, out=out)
    # what mrcal's projection uncertainty does with them (model_analysis.py:755-766): regularization present
    dF = rng.normal(size=(2, Nstate))
    Ax = F.solve_xt_JtJ_bt(dF)
    Var = amd._A_Jt_J_At__2(Ax, J.indptr, J.indices, J.data, Nleading_rows_J=Nlead)
    JtJ = (J.T @ J).toarray()
    Ad = np.linalg.solve(JtJ, dF.T).T
    assert np.abs(Var - Ad @ Jl.T @ Jl @ Ad.T).max() < 1e-7*np.abs(Var).max()


@pytest.mark.timeout(600)
def test_Jt_x_at_the_metric_size_is_exact_and_reproducible(amd):
    """y = Jt x at 1.6 M x 6140 with 37 M entries, and with a y longer than one LDS tile (16 cameras x 1400 frames:
    8684 columns, two tiles): against scipy, and the same bits three times"""
    for Ncameras, Nframes in ((8, 1000), (16, 1400)):
        oi, _ = make_calibration_problem(amd._api, Ncameras=Ncameras, Nframes=Nframes, lensmodel="LENSMODEL_OPENCV8",
                                         object_width_n=10, object_height_n=10, seed=0)
        _, x, J, _ = amd.optimizer_callback(no_factorization=True, **oi)
        ref = J.T @ x
        outs = []
        for rep in range(3):
            out = np.zeros((J.shape[1],))
            amd._Jt_x(J.indptr, J.indices, J.data, x, out=out)
            outs.append(out)
        assert np.abs(outs[0] - ref).max() < 1e-11*np.abs(ref).max()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_callback_factorization_is_the_problems_own_and_reproducible(amd):
    """optimizer_callback()'s factorization comes from the resident problem's atomics-free normal equations
    (mrcal_amd_factorization_create_from_problem): the same bits twice, and the same solve - to what two exact
    factorizations of a badly conditioned matrix leave of it - as the factorization built from the returned matrix"""
    oi, _ = make_calibration_problem(amd._api, Ncameras=3, Nframes=12, lensmodel="LENSMODEL_OPENCV8",
                                     object_width_n=8, object_height_n=7, seed=5)
    b, x, J, F  = amd.optimizer_callback(**oi)
    _, _, _, F2 = amd.optimizer_callback(**oi)
    assert F is not None and F2 is not None
    rng = np.random.RandomState(1)
    bt = rng.normal(size=(4, J.shape[1]))
    xt = F.solve_xt_JtJ_bt(bt)
    assert np.array_equal(xt, F2.solve_xt_JtJ_bt(bt))
    assert F.rcond() == F2.rcond()
    Fj = amd.CHOLMOD_factorization(J, _partition=None) if J.shape[1] < 400 else None
    Jd = J.toarray(); N = Jd.T @ Jd
    resid = np.abs(xt @ N - bt).max() / (np.abs(N).max()*np.abs(xt).max())
    assert resid < 1e-10
    # the products the uncertainty code takes from the resident J
    xx = rng.normal(size=(J.shape[0],))
    assert np.abs(F._Jt_x(xx) - J.T @ xx).max() < 1e-11*np.abs(J.T @ xx).max()
    assert np.array_equal(F._Jt_x(xx), F2._Jt_x(xx))
    # a state where JtJ is singular: None, like the reference (mrcal-pywrap.c:1981-1988)
    o2 = dict(oi, do_apply_regularization=False)
    o2["observations_board"] = oi["observations_board"].copy(); o2["observations_board"][..., 2] = -1.     # nothing observed
    assert amd.optimizer_callback(**o2)[3] is None


@pytest.mark.parametrize("case", ("boards", "dense", "wide range"))
def test_factorization_of_a_bare_matrix_is_reproducible(amd, case):
    """CHOLMOD_factorization(J) of a matrix the caller made: the block normal equations are summed row by row with
    atomics, in sums made so that no addition rounds (launch_assemble_rows: three levels of pre-rounded parts, the
    columns' largest entries setting the units) - the same bits whatever order the atomics land in - and exact: against
    numpy's JtJ, and against the plain sums of before (MRCAL_AMD_PLAIN_ROW_SUMS in a second process is not needed: the
    solve is the check). Until round 4 this was the last user-visible result that was not the same twice"""
    import scipy.sparse
    from mrcal_amd import CHOLMOD_factorization
    rng = np.random.RandomState(7)
    if case == "boards":
        oi, _ = make_calibration_problem(amd._api, Ncameras=3, Nframes=40, lensmodel="LENSMODEL_OPENCV8",
                                         object_width_n=10, object_height_n=10, seed=9)
        _, _, J, _ = amd.optimizer_callback(**oi, no_factorization=True)
        part = (3*12 + 2*6, 40, 0, 2)        # [intrinsics | extrinsics] leading, 40 frames, the board's warp last
        assert J.shape[1] == part[0] + 6*part[1] + part[3]
    else:
        Nc, Nfb, rows = 37, 25, []
        for f in range(Nfb):
            for _ in range(60):
                r = np.zeros(Nc + 6*Nfb); r[:Nc] = rng.normal(size=Nc)*(rng.uniform(size=Nc) < 0.4)
                r[Nc + 6*f: Nc + 6*f + 6] = rng.normal(size=6)
                rows.append(r)
        A = np.array(rows)
        if case == "wide range":
            A = A * 10.0**rng.uniform(-40, 40, size=A.shape[1])[None, :]      # columns from 1e-40 to 1e+40
        J, part = scipy.sparse.csr_matrix(A), (Nc, Nfb, 0, 0)
    Fs = [CHOLMOD_factorization(J, _partition=part) for _ in range(3)]
    bt = rng.normal(size=(3, J.shape[1]))
    xs = [F.solve_xt_JtJ_bt(bt) for F in Fs]
    for x in xs[1:]:
        assert np.array_equal(x, xs[0])
    assert Fs[0].rcond() == Fs[1].rcond() == Fs[2].rcond()
    if case != "wide range":
        Jd = J.toarray(); N = Jd.T @ Jd
        resid = np.abs(xs[0] @ N - bt).max() / (np.abs(N).max()*np.abs(xs[0]).max())
        assert resid < 1e-10, resid
    else:
        # (JtJ of such columns does not fit numpy's doubles' range in one matrix norm: column by column)
        Jd = J.toarray(); s = np.abs(Jd).max(axis=0)
        Ns = (Jd/s).T @ (Jd/s)
        resid = np.abs((xs[0]*s) @ Ns - bt/s).max() / np.abs(bt/s).max()
        assert resid < 1e-8, resid


def test_factorization_of_a_bare_matrix_with_a_column_listed_twice_in_a_run(amd):
    """CSR input may list a column twice in a row (scipy does not forbid it; the entries add). In a RUN of
    such rows - 8 or more with the same column list, which is what the half-wave path of launch_assemble_rows sums
    across lanes - the cross term of (v_p + v_q)^2 must land on the diagonal entry twice, as it does when rows go one
    by one. Runs of 40 rows with a repeated camera-block column and a repeated frame-block column, beside rows without
    runs: JtJ against numpy's from the summed-up dense matrix, three times the same bits"""
    import scipy.sparse
    from mrcal_amd import CHOLMOD_factorization
    rng = np.random.RandomState(11)
    Nc, Nfb = 9, 6
    Nstate = Nc + 6*Nfb
    indptr, indices, data = [0], [], []
    for f in range(Nfb):
        cols = [1, 4, 4, 7, Nc + 6*f + 0, Nc + 6*f + 2, Nc + 6*f + 2, Nc + 6*f + 5]       # 4 and one frame variable: twice
        for _ in range(40):                                                                  # a run: the same list, row after row
            indices += cols; data += list(rng.normal(size=len(cols))); indptr.append(len(indices))
        for _ in range(5):                                                                   # and rows of their own
            c = sorted(rng.choice(Nc, size=4, replace=False)) + [Nc + 6*f + k for k in range(6)]
            indices += c; data += list(rng.normal(size=len(c))); indptr.append(len(indices))
    J = scipy.sparse.csr_matrix((np.array(data), np.array(indices, dtype=np.int32), np.array(indptr, dtype=np.int32)),
                                shape=(len(indptr) - 1, Nstate))
    assert J.nnz == len(data) and not J.has_canonical_format
    Jd = np.zeros(J.shape)
    for r in range(J.shape[0]):
        for k in range(J.indptr[r], J.indptr[r+1]): Jd[r, J.indices[k]] += J.data[k]
    N = Jd.T @ Jd
    bt = rng.normal(size=(3, Nstate))
    xs = [CHOLMOD_factorization(J, _partition=(Nc, Nfb, 0, 0)).solve_xt_JtJ_bt(bt) for _ in range(3)]
    for x in xs[1:]: assert np.array_equal(x, xs[0])
    ref = np.linalg.solve(N, bt.T).T
    assert np.abs(xs[0] - ref).max() < 1e-9*np.abs(ref).max(), np.abs(xs[0] - ref).max()
