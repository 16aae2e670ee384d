"""An arbiter that is neither the product nor the checker's solver: is a returned calibration a stationary point of
the reference's cost? The residuals and the Jacobian come from an optimizer_callback() (the reference's own,
oracle/_ref, unless told otherwise); the judgement from numpy and from scipy.optimize.least_squares(method="trf",
x_scale="jac") started at the returned state: a solver that shares no code with libdogleg, its restatement or the
GPU solver must not be able to lower the cost from there. Test infrastructure."""
import numpy as np
from mrcal_amd.synthetic import copy_inputs


def with_state(api, oi, b_packed):
    """a copy of the inputs with the packed state b written into the arrays the callback reads"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import schur_numpy
    return schur_numpy.with_state(api, oi, b_packed, copy_inputs)


def stationarity(api, oi_solved):
    """(|Jt x| / (|J|_F |x|), cost, b_packed) at the state the arrays hold"""
    b, x, J, _ = api.optimizer_callback(no_factorization=True, **copy_inputs(oi_solved))
    g = J.T @ x
    return float(np.linalg.norm(g)/(np.sqrt((J.data**2).sum())*np.linalg.norm(x))), float(x @ x), b


def least_squares_gain(api, oi_solved, max_nfev=30):
    """relative decrease of the cost that scipy's trust-region-reflective solver finds from the returned state
    (outlier weights as returned: the arbiter solves the problem the last dog-leg pass solved). 0 = it found
    nothing"""
    import scipy.optimize
    oi = copy_inputs(oi_solved)
    oi["do_apply_outlier_rejection"] = False
    b0 = api.optimizer_callback(no_factorization=True, no_jacobian=True, **copy_inputs(oi))[0]
    def fun(b): return api.optimizer_callback(no_factorization=True, no_jacobian=True, **with_state(api, oi, b))[1]
    def jac(b): return api.optimizer_callback(no_factorization=True, **with_state(api, oi, b))[2]
    x0 = fun(b0)
    # (the round trip pack -> unpack -> pack of with_state() must not move the point)
    assert np.abs(x0 - api.optimizer_callback(no_factorization=True, no_jacobian=True, **copy_inputs(oi))[1]).max() < 1e-9
    r = scipy.optimize.least_squares(fun, b0, jac=jac, method="trf", x_scale="jac", max_nfev=max_nfev,
                                     ftol=1e-15, xtol=1e-15, gtol=1e-15)
    c0, c1 = float(x0 @ x0), float(2.0*r.cost)
    return (c0 - c1)/c0
