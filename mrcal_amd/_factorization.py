"""CHOLMOD_factorization-equivalent: factorize JtJ on the GPU, solve against it.

Reference: the mrcal.CHOLMOD_factorization type, mrcal-pywrap.c:111-214 (ctor:
cholmod_analyze + cholmod_factorize of Jt), :425-569 solve_xt_JtJ_bt(bt,sys),
:580-592 rcond(), and its test test/test-CHOLMOD-factorization.py.

    F  = mrcal_amd.CHOLMOD_factorization(J)      # J: scipy.sparse.csr_matrix
    xt = F.solve_xt_JtJ_bt(bt)                   # rows of bt are right-hand sides

optimizer_callback() returns one of these, built with the state partition of
the problem, so that the structured (Schur complement) solver of the
optimization is what runs. Constructed from a bare matrix, JtJ is treated as
dense: meant for small matrices. Only sys='A' is implemented; the
permuted-triangular-factor systems ('P','L','D',...) expose CHOLMOD's own
ordering and have no meaning here.
"""
import ctypes as C
import numpy as np


class CHOLMOD_factorization:
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
        L.mrcal_amd_factorization_destroy.restype  = None
        L.mrcal_amd_factorization_destroy.argtypes = [vp]
        L.mrcal_amd_factorization_solve.restype  = C.c_bool
        L.mrcal_amd_factorization_solve.argtypes = [vp, vp, C.c_int, vp]
        L.mrcal_amd_factorization_rcond.restype  = C.c_double
        L.mrcal_amd_factorization_rcond.argtypes = [vp]
        L._mrcal_amd_factorization_declared = True

    def solve_xt_JtJ_bt(self, bt, sys="A"):
        """xt such that (JtJ) x = b for every row b of bt (..., Nstate)"""
        if sys != "A":
            raise NotImplementedError(
                f"solve_xt_JtJ_bt(sys='{sys}'): only sys='A' is available. The other systems address CHOLMOD's "
                "permuted triangular factors, which this solver does not have")
        bt = np.asarray(bt)
        if bt.ndim < 1:
            raise RuntimeError(f"bt must be at least a 1-dimensional numpy array. Instead got {bt.ndim} dimensions")
        if bt.dtype != np.float64 or not bt.flags.c_contiguous:
            raise RuntimeError("bt must be a C-contiguous array of float64")
        if bt.shape[-1] != self._Nstate:
            raise RuntimeError(f"bt must have {self._Nstate} columns; got {bt.shape[-1]}")
        out = np.empty_like(bt)
        Nrhs = bt.size // self._Nstate if self._Nstate else 0
        if Nrhs and not self._L.mrcal_amd_factorization_solve(self._h, bt.ctypes.data, Nrhs, out.ctypes.data):
            raise RuntimeError("solve failed")
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
