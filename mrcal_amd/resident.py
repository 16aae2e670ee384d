"""Resident tier: a calibration problem held in HBM across evaluations and
solver steps (include/mrcal_amd.h, mrcal_amd_problem_*). This is what the
benchmark and the multi-GPU driver use; optimize()/optimizer_callback() are the
one-shot forms of the same thing.
"""
import ctypes as C
import numpy as np

from ._cabi import _ptr, Lensmodel, ProblemSelections


class Problem:
    """One optimization problem (or one frame-shard of it) resident on the
    current HIP device.

        p = Problem(**optimization_inputs)
        p.evaluate()                 # x, J at the resident packed state
        x = p.x()                    # host copies on demand
    """

    def __init__(self, _shard=(0,-1), _leader=True, _shard_points=(0,-1), _shard_tripoints=(0,-1), _ingested=None,
                 **optimization_inputs):
        from . import _api, _lib
        self._api = _api
        self._lib = _lib.lib
        self._declare()
        # (_ingested: the marshalled arguments of an optimizer_callback() call, which holds them already)
        p = _ingested if _ingested is not None else _api._ingest(optimization_inputs, callback=False)
        self._inputs = p   # keeps the numpy arrays alive
        a = _api._common_args(p)
        # common args: ..., lensmodel, imagersizes, sel, problem_constants, spacing, W, H, verbose
        self.handle = self._lib.mrcal_amd_problem_create_sharded(
            *a[:18], a[18], a[19], a[20], a[22], a[23], a[24],
            int(_shard[0]), int(_shard[1]), int(_shard_points[0]), int(_shard_points[1]),
            int(_shard_tripoints[0]), int(_shard_tripoints[1]), bool(_leader))
        if not self.handle:
            raise RuntimeError("mrcal_amd_problem_create() failed:" + _api._last_error())
        self.Nstate = self._lib.mrcal_amd_problem_Nstate(self.handle)
        self.Nmeas  = self._lib.mrcal_amd_problem_Nmeasurements(self.handle)
        self.Nnz    = self._lib.mrcal_amd_problem_Nnz(self.handle)
        self.Nstate_global, self.Nmeas_global, self.Nnz_global = self.Nstate, self.Nmeas, self.Nnz

    def _declare(self):
        L = self._lib
        if getattr(L, "_mrcal_amd_resident_declared", False):
            return
        vp = C.c_void_p
        L.mrcal_amd_problem_create_sharded.restype  = vp
        L.mrcal_amd_problem_create_sharded.argtypes = [
            vp, vp, vp, vp, vp,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            vp, vp, C.c_int, C.c_int,
            vp, C.c_int,
            vp, vp,
            C.POINTER(Lensmodel), vp, ProblemSelections,
            C.c_double, C.c_int, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_bool]
        L.mrcal_amd_problem_destroy.restype  = None
        L.mrcal_amd_problem_destroy.argtypes = [vp]
        for name in ("Nstate", "Nmeasurements"):
            f = getattr(L, f"mrcal_amd_problem_{name}")
            f.restype, f.argtypes = C.c_int, [vp]
        L.mrcal_amd_problem_Nnz.restype, L.mrcal_amd_problem_Nnz.argtypes = C.c_int64, [vp]
        L.mrcal_amd_problem_jacobian_algorithmic_bytes.restype  = C.c_int64
        L.mrcal_amd_problem_jacobian_algorithmic_bytes.argtypes = [vp]
        L.mrcal_amd_problem_synchronize.restype, L.mrcal_amd_problem_synchronize.argtypes = C.c_bool, [vp]
        for name in ("dev_b_packed", "dev_x", "dev_J_rowptr", "dev_J_colidx", "dev_J_values", "stream"):
            f = getattr(L, f"mrcal_amd_problem_{name}")
            f.restype, f.argtypes = vp, [vp]
        L.mrcal_amd_problem_set_b_packed.restype, L.mrcal_amd_problem_set_b_packed.argtypes = C.c_bool, [vp, vp]
        L.mrcal_amd_problem_get_b_packed.restype, L.mrcal_amd_problem_get_b_packed.argtypes = C.c_bool, [vp, vp]
        L.mrcal_amd_problem_get_x.restype,        L.mrcal_amd_problem_get_x.argtypes        = C.c_bool, [vp, vp]
        L.mrcal_amd_problem_get_J.restype,        L.mrcal_amd_problem_get_J.argtypes        = C.c_bool, [vp, vp, vp, vp]
        L.mrcal_amd_problem_evaluate.restype,     L.mrcal_amd_problem_evaluate.argtypes     = C.c_bool, [vp, C.c_bool, C.c_bool]
        L.mrcal_amd_problem_last_jacobian_kernel_ms.restype  = C.c_double
        L.mrcal_amd_problem_last_jacobian_kernel_ms.argtypes = [vp]
        ip = C.POINTER(C.c_int)
        dpp = C.POINTER(C.c_double)
        L.mrcal_amd_problem_solve.restype,     L.mrcal_amd_problem_solve.argtypes     = C.c_double, [vp, C.c_int, ip]
        L.mrcal_amd_problem_run_steps.restype, L.mrcal_amd_problem_run_steps.argtypes = C.c_int,    [vp, C.c_int, dpp]
        L.mrcal_amd_problem_solver_stats.restype  = None
        L.mrcal_amd_problem_solver_stats.argtypes = [vp, ip, ip, ip, ip, dpp, dpp, dpp]
        L.mrcal_amd_problem_get_normal_equations.restype  = C.c_bool
        L.mrcal_amd_problem_get_normal_equations.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.mrcal_amd_problem_gauss_newton_step.restype, L.mrcal_amd_problem_gauss_newton_step.argtypes = C.c_bool, [vp, vp]
        L.mrcal_amd_problem_get_board_pool.restype,    L.mrcal_amd_problem_get_board_pool.argtypes    = C.c_bool, [vp, vp]
        L.mrcal_amd_problem_jacobian_timing_begin.restype,  L.mrcal_amd_problem_jacobian_timing_begin.argtypes = C.c_bool, [vp, C.c_int]
        L.mrcal_amd_problem_jacobian_timing_begin_strided.restype,  L.mrcal_amd_problem_jacobian_timing_begin_strided.argtypes = C.c_bool, [vp, C.c_int, C.c_int]
        L.mrcal_amd_problem_jacobian_timing_end.restype  = C.c_bool
        L.mrcal_amd_problem_jacobian_timing_end.argtypes = [vp, ip, dpp, dpp, dpp]
        L.mrcal_amd_problem_dissection.restype, L.mrcal_amd_problem_dissection.argtypes = C.c_bool, [vp, ip]
        L.mrcal_amd_problem_set_jacobian_stream.restype, L.mrcal_amd_problem_set_jacobian_stream.argtypes = C.c_int, [vp, C.c_int]
        L.mrcal_amd_problem_jacobian_stream_is_optional.restype, L.mrcal_amd_problem_jacobian_stream_is_optional.argtypes = C.c_int, [vp]
        L._mrcal_amd_resident_declared = True

    def _check(self, ok, what):
        if not ok:
            raise RuntimeError(f"{what} failed:" + self._api._last_error())

    def close(self):
        if getattr(self, "handle", None):
            self._lib.mrcal_amd_problem_destroy(self.handle)
            self.handle = None
    def __del__(self):
        try:    self.close()
        except Exception: pass
    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def evaluate(self, with_jacobian=True, sync=True):
        self._check(self._lib.mrcal_amd_problem_evaluate(self.handle, with_jacobian, sync), "evaluate")

    def jacobian_kernel_ms(self):
        return self._lib.mrcal_amd_problem_last_jacobian_kernel_ms(self.handle)

    def synchronize(self):
        self._check(self._lib.mrcal_amd_problem_synchronize(self.handle), "synchronize")

    def jacobian_algorithmic_bytes(self):
        return int(self._lib.mrcal_amd_problem_jacobian_algorithmic_bytes(self.handle))

    def jacobian_timing_begin(self, capacity, stride=1):
        self._check(self._lib.mrcal_amd_problem_jacobian_timing_begin_strided(self.handle, int(capacity), int(stride)), "jacobian_timing_begin")

    def jacobian_timing_end(self):
        """(Nlaunches, total_ms, min_ms, max_ms) of the Jacobian kernel since _begin()"""
        n = C.c_int(0); t = C.c_double(0); mn = C.c_double(0); mx = C.c_double(0)
        self._check(self._lib.mrcal_amd_problem_jacobian_timing_end(self.handle, C.byref(n), C.byref(t), C.byref(mn), C.byref(mx)),
                    "jacobian_timing_end")
        return n.value, t.value, mn.value, mx.value

    def dissection(self):
        """the nested-dissection order of a splined problem's camera block: dict(rounds, ns_max, active, nA, nB, nS) -
        rounds == 0: not in use (include/mrcal_amd.h)"""
        out = (C.c_int*9)()
        self._check(self._lib.mrcal_amd_problem_dissection(self.handle, out), "dissection")
        return dict(zip(("rounds", "ns_max", "active", "nA", "nB", "nS", "ideal_A", "ideal_B", "ideal_S"), [int(v) for v in out]))

    def set_jacobian_stream(self, stream):
        """stream=False: solve() / run_steps() do not write the CSR values of J to HBM in every step (nothing in the
        solve reads them; J() evaluates them on demand afterwards). The same bits in every result. Returns the previous
        setting (include/mrcal_amd.h)"""
        return bool(self._lib.mrcal_amd_problem_set_jacobian_stream(self.handle, 1 if stream else 0))

    def jacobian_stream_is_optional(self):
        return bool(self._lib.mrcal_amd_problem_jacobian_stream_is_optional(self.handle))

    def set_b_packed(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert b.shape == (self.Nstate,)
        self._check(self._lib.mrcal_amd_problem_set_b_packed(self.handle, _ptr(b)), "set_b_packed")

    def b_packed(self):
        b = np.empty((self.Nstate,), dtype=np.float64)
        self._check(self._lib.mrcal_amd_problem_get_b_packed(self.handle, _ptr(b)), "get_b_packed")
        return b

    def x(self):
        x = np.empty((self.Nmeas,), dtype=np.float64)
        self._check(self._lib.mrcal_amd_problem_get_x(self.handle, _ptr(x)), "get_x")
        return x

    def J(self):
        import scipy.sparse
        P = np.empty((self.Nmeas+1,), dtype=np.int32)
        I = np.empty((self.Nnz,),     dtype=np.int32)
        X = np.empty((self.Nnz,),     dtype=np.float64)
        self._check(self._lib.mrcal_amd_problem_get_J(self.handle, _ptr(P), _ptr(I), _ptr(X)), "get_J")
        return scipy.sparse.csr_matrix((X, I, P), shape=(self.Nmeas, self.Nstate))

    def stream(self):
        return self._lib.mrcal_amd_problem_stream(self.handle)

    # ---------------------------------------------------------------- solver
    def solve(self, max_iterations=0):
        """the whole solve (dog leg + outlier rejection); returns a stats dict.
        The solution stays resident: b_packed(), x()"""
        nout = C.c_int(0)
        rms = self._lib.mrcal_amd_problem_solve(self.handle, int(max_iterations), C.byref(nout))
        if rms < 0:
            raise RuntimeError("mrcal_amd_problem_solve() failed:" + self._api._last_error())
        st = self.solver_stats()
        st.update(rms_reproj_error__pixels=rms, Noutliers_board=nout.value)
        return st

    def run_steps(self, Nsteps, trustregion=None):
        """exactly Nsteps dog-leg steps; returns (steps done, trust region)"""
        tr = C.c_double(-1.0 if trustregion is None else float(trustregion))
        n = self._lib.mrcal_amd_problem_run_steps(self.handle, int(Nsteps), C.byref(tr))
        if n < 0:
            raise RuntimeError("mrcal_amd_problem_run_steps() failed:" + self._api._last_error())
        return n, tr.value

    def solver_stats(self):
        i = [C.c_int(0) for _ in range(4)]
        d = [C.c_double(0) for _ in range(3)]
        self._lib.mrcal_amd_problem_solver_stats(self.handle, *[C.byref(v) for v in i], *[C.byref(v) for v in d])
        return dict(Niterations=i[0].value, Nevaluations=i[1].value, Nfactorizations=i[2].value,
                    Noutlier_passes=i[3].value, norm2_x=d[0].value, lambda_=d[1].value, seconds=d[2].value)

    def factorization(self):
        """CHOLMOD_factorization of JtJ at the resident state (None if singular), from this problem's own normal
        equations: no atomics anywhere (include/mrcal_amd.h, mrcal_amd_factorization_create_from_problem)"""
        from ._factorization import CHOLMOD_factorization
        return CHOLMOD_factorization._from_problem(self)

    def normal_equations(self):
        """evaluates at the resident state; returns dict(A,Bt,D,g,norm2_x,+dims). Nie is S_split (== the partition()'s):
        the number of intrinsics + extrinsics variables when the frames are eliminated, of intrinsics variables alone
        when the extrinsics are (include/mrcal_amd.h)"""
        dims = (C.c_int*6)()
        self._check(self._lib.mrcal_amd_problem_get_normal_equations(self.handle, None,None,None,None,None, dims),
                    "get_normal_equations")
        Nc, NE, NEb, Nfb, Nie, Nwarp = list(dims)
        A  = np.zeros((Nc,Nc)); Bt = np.zeros((NE,Nc)); D = np.zeros((NEb,6,6)); g = np.zeros((self.Nstate,))
        n2 = np.zeros((1,))
        self._check(self._lib.mrcal_amd_problem_get_normal_equations(self.handle, _ptr(A), _ptr(Bt), _ptr(D), _ptr(g), _ptr(n2), dims),
                    "get_normal_equations")
        return dict(A=A, Bt=Bt, D=D, g=g, norm2_x=n2[0], Nc=Nc, NE=NE, NEb=NEb, Nfb=Nfb, Nie=Nie, Nwarp=Nwarp, **self.partition())

    def partition(self):
        """dict(S_split, S_shift, E_state0, eliminates): S index s is state s (s < S_split) or s + S_shift, E index
        e is state E_state0 + e; eliminates is 'frames' or 'extrinsics' (include/mrcal_amd.h)"""
        info = (C.c_int*4)()
        f = self._lib.mrcal_amd_problem_partition
        f.restype, f.argtypes = None, [C.c_void_p, C.POINTER(C.c_int)]
        f(self.handle, info)
        return dict(S_split=info[0], S_shift=info[1], E_state0=info[2], eliminates="extrinsics" if info[3] else "frames")

    def drt_cross_reprojection__dbpacked(self, icam_intrinsics=-1):
        """K (6,Nstate) of mrcal.drt_cross_reprojection__dbpacked() from the RESIDENT Jacobian at the current
        state: J never leaves the device"""
        f = self._lib.mrcal_amd_problem_drt_cross_reprojection
        f.restype, f.argtypes = C.c_bool, [C.c_void_p, C.c_int, C.c_void_p]
        K = np.zeros((6, self.Nstate))
        self._check(f(self.handle, -1 if icam_intrinsics is None else int(icam_intrinsics), _ptr(K)), "drt_cross_reprojection")
        return K

    def gauss_newton_step(self):
        d = np.zeros((self.Nstate,))
        self._check(self._lib.mrcal_amd_problem_gauss_newton_step(self.handle, _ptr(d)), "gauss_newton_step")
        return d

    def lchol_diag_ratio(self):
        """min / max of the diagonal of the big camera block's Cholesky factors over the last dog-leg pass; 1.0 where no
        such factorization ran (include/mrcal_amd.h)"""
        f = self._lib.mrcal_amd_problem_lchol_diag_ratio
        f.restype, f.argtypes = C.c_double, [C.c_void_p]
        return float(f(self.handle))

    def uses_sweep(self):
        """has this problem's big Cholesky gone over to the backward sweep (the automatic stable fallback, or the test hook)?"""
        f = self._lib.mrcal_amd_problem_uses_sweep
        f.restype, f.argtypes = C.c_int, [C.c_void_p]
        return bool(f(self.handle))
