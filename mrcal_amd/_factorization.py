"""CHOLMOD_factorization-equivalent: factorize JtJ, solve against it.

Reference: the mrcal.CHOLMOD_factorization type, mrcal-pywrap.c:111-214 (ctor:
cholmod_analyze + cholmod_factorize of Jt), :425-569 solve_xt_JtJ_bt(bt,sys),
:580-592 rcond().
"""
import numpy as np


class CHOLMOD_factorization:
    def __init__(self, J=None):
        import scipy.sparse
        if J is None:
            raise RuntimeError("A CHOLMOD_factorization must be constructed from a Jacobian")
        if not scipy.sparse.isspmatrix_csr(J):
            raise RuntimeError("J must be a scipy.sparse.csr_matrix")
        self._J = J
        raise NotImplementedError("CHOLMOD_factorization: GPU factorization is not wired up yet")

    def solve_xt_JtJ_bt(self, bt, sys="A"):
        raise NotImplementedError

    def rcond(self):
        raise NotImplementedError
