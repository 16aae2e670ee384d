"""Pose arithmetic on the host, numpy only: the small subset of mrcal's poseutils
(reference: mrcal/poseutils.py, poseutils.c; conventions documented in
doc/conventions.org) that the camera-model files and the seeding need.

  rt = (r, t): a Rodrigues rotation r (3,) and a translation t (3,), shape (...,6)
  Rt = [R; t]: a rotation matrix R (3,3) stacked on a translation row, shape (...,4,3)
  a transform takes a point x to R x + t

Values only (no gradients): the gradients the solver needs live in the kernels
(device_math.hpp). Everything broadcasts over leading dimensions.
"""
import numpy as np


def identity_R():  return np.eye(3)
def identity_r():  return np.zeros(3)
def identity_Rt(): return np.vstack((np.eye(3), np.zeros(3)))
def identity_rt(): return np.zeros(6)


def R_from_r(r):
    """Rodrigues vector(s) (...,3) -> rotation matrices (...,3,3)"""
    r = np.asarray(r, dtype=float)
    th2 = np.sum(r*r, axis=-1)[..., None, None]
    th = np.sqrt(th2)
    K = np.zeros(r.shape[:-1] + (3, 3))
    K[..., 0, 1] = -r[..., 2]; K[..., 0, 2] =  r[..., 1]
    K[..., 1, 0] =  r[..., 2]; K[..., 1, 2] = -r[..., 0]
    K[..., 2, 0] = -r[..., 1]; K[..., 2, 1] =  r[..., 0]
    small = th2 < 1e-12
    th_safe = np.where(small, 1.0, th)
    a = np.where(small, 1.0 - th2/6.0, np.sin(th_safe)/th_safe)                    # sin(th)/th
    b = np.where(small, 0.5 - th2/24.0, (1.0 - np.cos(th_safe))/(th_safe*th_safe)) # (1-cos(th))/th^2
    return np.eye(3) + a*K + b*(K @ K)


def r_from_R(R):
    """rotation matrices (...,3,3) -> Rodrigues vectors (...,3), |r| in [0,pi]"""
    R = np.asarray(R, dtype=float)
    lead = R.shape[:-2]
    Rf = R.reshape((-1, 3, 3))
    out = np.empty((Rf.shape[0], 3))
    for i, M in enumerate(Rf):
        v = np.array((M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]))   # 2 sin(th) axis
        s = 0.5*np.linalg.norm(v)
        c = 0.5*(np.trace(M) - 1.0)
        th = np.arctan2(s, c)
        if c > 0 and s <= 1e-6:
            out[i] = 0.5*v                      # th ~ sin(th)
        elif c > 0 or s > 0.1:
            out[i] = v*(th/(2.0*s))
        else:
            # towards pi v = 2 sin(th) axis shrinks and its rounding error (1e-16, absolute) does not:
            # at pi - 1e-5 the axis from v is good to 1e-11 only
            # near pi: the axis from the symmetric part, sym(R) - cos(th) I = (1 - cos(th)) a a^T
            B = 0.5*(M + M.T) - c*np.eye(3)
            a = B[int(np.argmax(np.diag(B)))]
            a = a/np.linalg.norm(a)
            if np.dot(a, v) < 0: a = -a
            out[i] = a*th
    return out.reshape(lead + (3,))


def Rt_from_rt(rt):
    rt = np.asarray(rt, dtype=float)
    return np.concatenate((R_from_r(rt[..., :3]), rt[..., None, 3:]), axis=-2)


def rt_from_Rt(Rt):
    Rt = np.asarray(Rt, dtype=float)
    return np.concatenate((r_from_R(Rt[..., :3, :]), Rt[..., 3, :]), axis=-1)


def invert_R(R):
    return np.swapaxes(np.asarray(R, dtype=float), -1, -2)


def invert_Rt(Rt):
    """x = R^T (y - t)"""
    Rt = np.asarray(Rt, dtype=float)
    Rinv = np.swapaxes(Rt[..., :3, :], -1, -2)
    tinv = -np.einsum("...ij,...j->...i", Rinv, Rt[..., 3, :])
    return np.concatenate((Rinv, tinv[..., None, :]), axis=-2)


def invert_rt(rt):
    rt = np.asarray(rt, dtype=float)
    r = -rt[..., :3]
    t = -np.einsum("...ij,...j->...i", R_from_r(r), rt[..., 3:])
    return np.concatenate((r, t), axis=-1)


def compose_Rt(*Rts):
    """compose_Rt(A,B,C) applies C first: x -> A(B(C(x)))"""
    out = np.asarray(Rts[0], dtype=float)
    for Rt in Rts[1:]:
        Rt = np.asarray(Rt, dtype=float)
        R = out[..., :3, :] @ Rt[..., :3, :]
        t = np.einsum("...ij,...j->...i", out[..., :3, :], Rt[..., 3, :]) + out[..., 3, :]
        out = np.concatenate((R, t[..., None, :]), axis=-2)
    return out


def compose_rt(*rts):
    return rt_from_Rt(compose_Rt(*[Rt_from_rt(rt) for rt in rts]))


def compose_r(*rs):
    R = R_from_r(rs[0])
    for r in rs[1:]: R = R @ R_from_r(r)
    return r_from_R(R)


def rotate_point_R(R, x):
    return np.einsum("...ij,...j->...i", np.asarray(R, dtype=float), np.asarray(x, dtype=float))


def rotate_point_r(r, x):
    return rotate_point_R(R_from_r(r), x)


def transform_point_Rt(Rt, x):
    Rt = np.asarray(Rt, dtype=float)
    return rotate_point_R(Rt[..., :3, :], x) + Rt[..., 3, :]


def transform_point_rt(rt, x):
    return transform_point_Rt(Rt_from_rt(rt), x)


def close_contour(c):
    """a polygon (N,2) with its first point repeated at the end (reference:
    mrcal/utils.py close_contour); None and empty arrays pass through"""
    if c is None: return None
    c = np.asarray(c)
    if c.size == 0 or np.linalg.norm(c[0] - c[-1]) < 1e-6: return c
    return np.concatenate((c, c[:1]), axis=0)
