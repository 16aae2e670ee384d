"""An arbiter that is neither the product nor the checker's solver: is a returned calibration a stationary point of
the reference's cost? The residuals and the Jacobian come from an optimizer_callback() (the reference's own,
oracle/_ref, unless told otherwise); the judgement from numpy and from scipy.optimize.least_squares(method="trf",
x_scale="jac") started at the returned state: a solver that shares no code with libdogleg, its restatement or the
GPU solver must not be able to lower the cost from there. Test infrastructure."""
import numpy as np
from mrcal_amd.synthetic import copy_inputs


def with_state(api, oi, b_packed):
    """a copy of the inputs with the packed state b written into the arrays the callback reads"""
    o = copy_inputs(oi)
    b = np.array(b_packed, dtype=float)
    api.unpack_state(b, **o)
    core  = bool(o.get("do_optimize_intrinsics_core", True))
    dist  = bool(o.get("do_optimize_intrinsics_distortions", True))
    Ncam  = o["intrinsics"].shape[0]
    if core or dist:
        Nopt = api.num_intrinsics_optimization_params(**o)
        for i in range(Ncam):
            i0 = api.state_index_intrinsics(i, **o)
            if i0 is None: continue
            dst = o["intrinsics"][i, (0 if core else 4):(None if dist else 4)]
            assert dst.size == Nopt
            dst[:] = b[i0:i0+Nopt]
    if o.get("do_optimize_extrinsics", True):
        for i in range(o["rt_cam_ref"].shape[0]):
            i0 = api.state_index_extrinsics(i, **o)
            if i0 is not None: o["rt_cam_ref"].reshape(-1,6)[i] = b[i0:i0+6]
    if o.get("do_optimize_frames", True):
        for i in range(o["rt_ref_frame"].shape[0]):
            i0 = api.state_index_frames(i, **o)
            if i0 is not None: o["rt_ref_frame"].reshape(-1,6)[i] = b[i0:i0+6]
    if o.get("do_optimize_calobject_warp", False) and o.get("calobject_warp") is not None:
        i0 = api.state_index_calobject_warp(**o)
        if i0 is not None: o["calobject_warp"][:] = b[i0:i0+2]
    return o


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
