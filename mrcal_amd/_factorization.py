"""CHOLMOD_factorization-equivalent: factorize JtJ on the GPU, solve against it.

Reference: the mrcal.CHOLMOD_factorization type, mrcal-pywrap.c:111-214 (ctor:
cholmod_analyze + cholmod_factorize of Jt), :425-569 solve_xt_JtJ_bt(bt,sys),
:580-592 rcond(), and its test test/test-CHOLMOD-factorization.py.

    F  = mrcal_amd.CHOLMOD_factorization(J)      # J: scipy.sparse.csr_matrix
    xt = F.solve_xt_JtJ_bt(bt)                   # rows of bt are right-hand sides

optimizer_callback() returns one of these, built with the state partition of
the problem, so that the structured (Schur complement) solver of the
optimization is what runs. Constructed from a bare matrix, JtJ is treated as
dense: meant for small matrices.

sys= (mrcal-pywrap.c:467-493): all of CHOLMOD's systems. The factorization kept
here is  L L^T = P (JtJ) P^T  with P the ordering [frame blocks | point blocks |
intrinsics, extrinsics | warp] (the eliminated blocks first) and D = I, so
'LD' == 'L', 'DLt' == 'Lt' and 'D' copies. As with CHOLMOD, the vectors of the
L/D systems live in the order of P: 'P' takes a vector there, 'Pt' back. The
sequence mrcal's projection uncertainty runs (mrcal/model_analysis.py:837-843),
    A1 = F.solve_xt_JtJ_bt(b,  sys='P')
    A2 = F.solve_xt_JtJ_bt(A1, sys='L')
    A3 = F.solve_xt_JtJ_bt(A2, sys='D')
    Var = A2 A3^T  ==  b (JtJ)^-1 b^T
works unchanged. Also here, because the J the factorization was made from is
resident on the device: _Jt_x() and _A_Jt_J_At() (mrcal-genpywrap.py:477-731).
"""
import ctypes as C
import numpy as np


class CHOLMOD_factorization:
    @classmethod
    def _from_problem(cls, problem):
        """the factorization of JtJ at a resident problem's current state, from the problem's own (atomics-free) normal
        equations: what optimizer_callback() returns. None if JtJ is singular"""
        from . import _lib
        L = _lib.lib
        cls._declare(L)
        h = L.mrcal_amd_factorization_create_from_problem(problem.handle)
        if not h:
            return None
        self = cls.__new__(cls)
        self._L = L
        self._h = h
        self._Nstate = int(L.mrcal_amd_factorization_Nstate(h))
        self._Nmeas  = int(L.mrcal_amd_factorization_Nmeasurements(h))
        return self

    def __init__(self, J=None, _partition=None):
        import scipy.sparse
        from . import _lib
        if J is None:
            raise RuntimeError("A CHOLMOD_factorization must be constructed from a Jacobian")
        if not scipy.sparse.isspmatrix_csr(J):
            raise RuntimeError("J must be a scipy.sparse.csr_matrix")
        L = _lib.lib
        self._L = L
        self._declare(L)
        Nmeas, Nstate = J.shape
        P = np.ascontiguousarray(J.indptr,  dtype=np.int32)
        I = np.ascontiguousarray(J.indices, dtype=np.int32)
        X = np.ascontiguousarray(J.data,    dtype=np.float64)
        if _partition is None:
            _partition = (Nstate, 0, 0, 0)
        self._Nstate = Nstate
        self._Nmeas  = Nmeas
        self._h = L.mrcal_amd_factorization_create(
            Nmeas, Nstate, P.ctypes.data, I.ctypes.data, X.ctypes.data, *[int(v) for v in _partition])
        if not self._h:
            f = L.mrcal_amd_last_error
            f.restype = C.c_char_p
            raise RuntimeError("CHOLMOD_factorization: " + (f() or b"failed").decode())

    @staticmethod
    def _declare(L):
        if getattr(L, "_mrcal_amd_factorization_declared", False):
            return
        vp = C.c_void_p
        L.mrcal_amd_factorization_create.restype  = vp
        L.mrcal_amd_factorization_create.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mrcal_amd_factorization_create_from_problem.restype  = vp
        L.mrcal_amd_factorization_create_from_problem.argtypes = [vp]
        L.mrcal_amd_factorization_Nstate.restype, L.mrcal_amd_factorization_Nstate.argtypes = C.c_int, [vp]
        L.mrcal_amd_factorization_Nmeasurements.restype, L.mrcal_amd_factorization_Nmeasurements.argtypes = C.c_int, [vp]
        L.mrcal_amd_factorization_destroy.restype  = None
        L.mrcal_amd_factorization_destroy.argtypes = [vp]
        L.mrcal_amd_factorization_solve.restype  = C.c_bool
        L.mrcal_amd_factorization_solve.argtypes = [vp, vp, C.c_int, vp]
        L.mrcal_amd_factorization_rcond.restype  = C.c_double
        L.mrcal_amd_factorization_rcond.argtypes = [vp]
        L.mrcal_amd_factorization_solve_sys.restype  = C.c_bool
        L.mrcal_amd_factorization_solve_sys.argtypes = [vp, C.c_int, vp, C.c_int, vp]
        L.mrcal_amd_factorization_Jt_x.restype  = C.c_bool
        L.mrcal_amd_factorization_Jt_x.argtypes = [vp, vp, vp]
        L.mrcal_amd_factorization_A_Jt_J_At.restype  = C.c_bool
        L.mrcal_amd_factorization_A_Jt_J_At.argtypes = [vp, vp, C.c_int, C.c_int, vp]
        L.mrcal_amd_csr_Jt_x.restype  = C.c_bool
        L.mrcal_amd_csr_Jt_x.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.mrcal_amd_csr_A_Jt_J_At.restype  = C.c_bool
        L.mrcal_amd_csr_A_Jt_J_At.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp]
        L._mrcal_amd_factorization_declared = True

    # CHOLMOD's system codes (cholmod_core.h), which mrcal_amd_factorization_solve_sys() takes
    _SYS = dict(A=0, LDLt=1, LD=2, DLt=3, L=4, Lt=5, D=6, P=7, Pt=8)

    def solve_xt_JtJ_bt(self, bt, sys="A"):
        """xt such that (JtJ) x = b for every row b of bt (..., Nstate); or one
        of the other systems of cholmod_solve2() (module docstring)"""
        key = sys[8:] if isinstance(sys, str) and sys.startswith("CHOLMOD_") else sys
        if key not in self._SYS:
            raise RuntimeError(f"Unknown sys '{sys}' given. Known values of sys: (" + ",".join(self._SYS) + ",)")
        bt = np.asarray(bt)
        if bt.ndim < 1:
            raise RuntimeError(f"bt must be at least a 1-dimensional numpy array. Instead got {bt.ndim} dimensions")
        if bt.dtype != np.float64:
            raise RuntimeError("bt must have dtype=float")
        if not bt.flags.c_contiguous:
            raise RuntimeError("bt must live in contiguous memory")
        if bt.shape[-1] != self._Nstate:
            raise RuntimeError(f"bt must be a 2-dimensional numpy array with {self._Nstate} cols (that's what the "
                               f"factorization has). Instead got {bt.shape[-1]} cols")
        Nrhs = bt.size // self._Nstate if self._Nstate else 0
        if Nrhs == 0:
            return bt                       # degenerate input: returned as it is, like the reference
        out = np.empty_like(bt)
        if key == "A":
            ok = self._L.mrcal_amd_factorization_solve(self._h, bt.ctypes.data, Nrhs, out.ctypes.data)
        else:
            ok = self._L.mrcal_amd_factorization_solve_sys(self._h, self._SYS[key], bt.ctypes.data, Nrhs, out.ctypes.data)
        if not ok:
            raise RuntimeError("solve failed")
        return out

    def _Jt_x(self, xt):
        """Jt x with the resident J (mrcal._mrcal_npsp._Jt_x: mrcal-genpywrap.py:658-731)"""
        xt = np.ascontiguousarray(xt, dtype=np.float64)
        if xt.shape != (self._Nmeas,):
            raise RuntimeError("len(xt) must match the number of rows in J")
        out = np.empty((self._Nstate,), dtype=np.float64)
        if not self._L.mrcal_amd_factorization_Jt_x(self._h, xt.ctypes.data, out.ctypes.data):
            raise RuntimeError("_Jt_x failed:" + _last_error(self._L))
        return out

    def _A_Jt_J_At(self, A, Nleading_rows_J=-1):
        """matmult(A,Jt,J,At) over the leading rows of the resident J
        (mrcal._mrcal_npsp._A_Jt_J_At: mrcal-genpywrap.py:477-657). A: (...,Nx,Nstate)"""
        A = np.ascontiguousarray(A, dtype=np.float64)
        if A.ndim < 2 or A.shape[-1] != self._Nstate:
            raise RuntimeError(f"A must have shape (...,Nx,{self._Nstate})")
        Nx = A.shape[-2]
        out = np.empty(A.shape[:-2] + (Nx, Nx), dtype=np.float64)
        Af, of = A.reshape(-1, Nx, self._Nstate), out.reshape(-1, Nx, Nx)
        for k in range(Af.shape[0]):
            a = np.ascontiguousarray(Af[k]); o = np.empty((Nx, Nx))
            if not self._L.mrcal_amd_factorization_A_Jt_J_At(self._h, a.ctypes.data, Nx, int(Nleading_rows_J), o.ctypes.data):
                raise RuntimeError(_last_error(self._L).strip() or "_A_Jt_J_At failed")
            of[k] = o
        return out

    def rcond(self):
        return self._L.mrcal_amd_factorization_rcond(self._h)

    def __str__(self):
        return f"CHOLMOD_factorization-equivalent (GPU, structured): Nstate={self._Nstate}"

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.mrcal_amd_factorization_destroy(h)
            self._h = None


def _last_error(L):
    f = L.mrcal_amd_last_error
    f.restype = C.c_char_p
    return (f() or b"").decode()


def _csr_args(Jp, Ji, Jx):
    Jp = np.ascontiguousarray(Jp, dtype=np.int32); Ji = np.ascontiguousarray(Ji, dtype=np.int32)
    Jx = np.ascontiguousarray(Jx, dtype=np.float64)
    return Jp, Ji, Jx


def _Jt_x(Jp, Ji, Jx, xt, out=None):
    """mrcal._mrcal_npsp._Jt_x (mrcal-genpywrap.py:658-731): Jt x for a CSR J given
    as its indptr, indices, data; out (Nstate,) MUST be given (its length is the
    number of columns). On the GPU"""
    from . import _lib
    L = _lib.lib
    CHOLMOD_factorization._declare(L)
    if out is None:
        raise RuntimeError("The output array MUST be passed-in because there's no way to know its shape beforehand")
    Jp, Ji, Jx = _csr_args(Jp, Ji, Jx)
    xt = np.ascontiguousarray(xt, dtype=np.float64)
    if xt.shape != (len(Jp) - 1,):
        raise RuntimeError("len(xt) must match the number of rows in J")
    if out.dtype != np.float64 or not out.flags.c_contiguous:
        raise RuntimeError("out must be a contiguous array of float64")
    if not L.mrcal_amd_csr_Jt_x(len(Jp) - 1, out.shape[0], Jp.ctypes.data, Ji.ctypes.data, Jx.ctypes.data,
                                xt.ctypes.data, out.ctypes.data):
        raise RuntimeError("_Jt_x failed:" + _last_error(L))
    return out


def _A_Jt_J_At(A, Jp, Ji, Jx, Nleading_rows_J=-1, out=None):
    """mrcal._mrcal_npsp._A_Jt_J_At (mrcal-genpywrap.py:477-567): matmult(A,Jt,J,At)
    over the Nleading_rows_J leading rows of the CSR J. A: (Nx,Nstate). On the GPU"""
    from . import _lib
    L = _lib.lib
    CHOLMOD_factorization._declare(L)
    if Nleading_rows_J <= 0:
        raise RuntimeError("Nleading_rows_J must be passed, and must be > 0")
    Jp, Ji, Jx = _csr_args(Jp, Ji, Jx)
    A = np.ascontiguousarray(A, dtype=np.float64)
    if A.ndim != 2:
        raise RuntimeError("A must have shape (Nx,Nstate)")
    Nx, Nstate = A.shape
    if out is None: out = np.empty((Nx, Nx), dtype=np.float64)
    if not L.mrcal_amd_csr_A_Jt_J_At(len(Jp) - 1, Nstate, Jp.ctypes.data, Ji.ctypes.data, Jx.ctypes.data,
                                     A.ctypes.data, Nx, int(Nleading_rows_J), out.ctypes.data):
        raise RuntimeError(_last_error(L).strip() or "_A_Jt_J_At failed")
    return out


def _A_Jt_J_At__2(A, Jp, Ji, Jx, Nleading_rows_J=-1, out=None):
    """the same for A of shape (2,Nstate) (mrcal-genpywrap.py:569-656)"""
    A = np.asarray(A)
    if A.shape[0] != 2:
        raise RuntimeError("A must have shape (2,Nstate)")
    return _A_Jt_J_At(A, Jp, Ji, Jx, Nleading_rows_J=Nleading_rows_J, out=out)
