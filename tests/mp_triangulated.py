"""TEST INFRASTRUCTURE: a third opinion on the triangulated-pair residual.

The residual of a pair (mrcal.c:5180-5506 + triangulation.cc:958-1123) is
2 sqrt(2 - 2 cos th): for th -> 0 the subtraction cancels, and ANY double
implementation of that formula carries an absolute error of ~eps/th in the
residual and a relative error of ~eps/th^2 in its gradient. To tell "the two
implementations differ because of a bug" from "both are inside the formula's own
rounding noise", this module evaluates the same mathematical function of the
pair's 12 extrinsics variables in 60-digit arithmetic (mpmath), with forward-mode
derivatives carried in the same arithmetic, from the same double inputs.

Only the tests import this.
"""
import numpy as np
import mpmath as mp

mp.mp.dps = 60
EPS = 2.0**-52

# mrcal.c:60-76
SCALE_ROTATION_CAMERA    = 0.1*np.pi/180.0
SCALE_TRANSLATION_CAMERA = 1.0


class D:
    """value + gradient wrt N variables, all mpf"""
    __slots__ = ("x", "g")
    N = 12

    def __init__(self, x, g=None):
        self.x = mp.mpf(x)
        self.g = g if g is not None else [mp.mpf(0)]*D.N

    @staticmethod
    def var(x, i):
        g = [mp.mpf(0)]*D.N
        g[i] = mp.mpf(1)
        return D(x, g)

    @staticmethod
    def lift(a):
        return a if isinstance(a, D) else D(a)

    def __add__(a, b):
        b = D.lift(b); return D(a.x+b.x, [p+q for p, q in zip(a.g, b.g)])
    __radd__ = __add__
    def __sub__(a, b):
        b = D.lift(b); return D(a.x-b.x, [p-q for p, q in zip(a.g, b.g)])
    def __rsub__(a, b):
        return D.lift(b) - a
    def __neg__(a):
        return D(-a.x, [-p for p in a.g])
    def __mul__(a, b):
        b = D.lift(b); return D(a.x*b.x, [p*b.x + a.x*q for p, q in zip(a.g, b.g)])
    __rmul__ = __mul__
    def __truediv__(a, b):
        b = D.lift(b)
        inv = 1/b.x
        v = a.x*inv
        return D(v, [(p - v*q)*inv for p, q in zip(a.g, b.g)])
    def __rtruediv__(a, b):
        return D.lift(b)/a


def dsqrt(a):
    s = mp.sqrt(a.x)
    return D(s, [p/(2*s) for p in a.g])
def dsin(a):
    c = mp.cos(a.x); return D(mp.sin(a.x), [p*c for p in a.g])
def dcos(a):
    s = -mp.sin(a.x); return D(mp.cos(a.x), [p*s for p in a.g])


def dot(a, b):   return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]
def cross(a, b): return [a[1]*b[2]-a[2]*b[1], a[2]*b[0]-a[0]*b[2], a[0]*b[1]-a[1]*b[0]]


def rotate(r, x, inverse):
    """Rodrigues rotation R(r) x (R(r)^T x if inverse): poseutils.c
    mrcal_rotate_point_r"""
    if inverse: r = [-c for c in r]
    th2 = dot(r, r)
    if th2.x == 0:
        # first order: x + r cross x
        c = cross(r, x)
        return [x[i] + c[i] for i in range(3)]
    th = dsqrt(th2)
    k  = [c/th for c in r]
    s, c = dsin(th), dcos(th)
    kx = cross(k, x)
    kd = dot(k, x)
    return [x[i]*c + kx[i]*s + k[i]*kd*(1 - c) for i in range(3)]


def angle_small(v0, v1):
    """triangulation.cc:767-805"""
    costh = dot(v0, v1)/dsqrt(dot(v0, v0)*dot(v1, v1))
    if costh.x < 0: costh = -costh
    th_sq = costh*(-2) + 2
    if th_sq.x < mp.mpf("1e-21"): return D(0)
    return dsqrt(th_sq)


def sigmoid(x, knee):
    """triangulation.cc:899-953"""
    if x.x <= 0:    return D(0)
    if knee <= x.x: return D(1)
    b, c = mp.mpf(2)/knee, mp.mpf(1)/2
    a = (mp.mpf(2) if x.x < mp.mpf(knee)/2 else mp.mpf(-2))/knee/knee
    dx = x - mp.mpf(knee)/2
    return dx*(dx*a + b) + c


def tri_error(v0, v1, t01):
    """triangulation.cc:958-1123 (the 'new method'), chirality :576-638"""
    def cn2(a, b):
        c = cross(a, b); return dot(c, c)
    pr = 1/cn2(v0, v1)
    l0 = dsqrt(cn2(v1, t01)*pr)
    l1 = dsqrt(cn2(v0, t01)*pr)
    m  = [(v0[i]*l0 + t01[i] + v1[i]*l1)/2 for i in range(3)]
    err = angle_small(v0, m)*2
    w0 = D(0); w1 = D(0); w01 = D(0)
    for i in range(3):
        xn  = ( l1*v1[i] + t01[i]) - l0*v0[i]
        x0  = ( l1*v1[i] + t01[i]) + l0*v0[i]
        x1  = (-(l1*v1[i]) + t01[i]) - l0*v0[i]
        x01 = (-(l1*v1[i]) + t01[i]) + l0*v0[i]
        w0  = w0  + (x0 *x0  - xn*xn)
        w1  = w1  + (x1 *x1  - xn*xn)
        w01 = w01 + (x01*x01 - xn*xn)
    convergent = w0.x > 0 and w1.x > 0 and w01.x > 0
    if not convergent:
        evp = angle_small(v0, v1)
        err = err + evp*(sigmoid(-w0, 3) + sigmoid(-w1, 3) + sigmoid(-w01, 3))
    return err, convergent


def pair_error(v0, v1, rt0, rt1):
    """The residual of the pair (observation vector v0 in camera 0, v1 in
    camera 1; rt = rt_cam_ref or None for the camera at the reference) and its
    derivatives wrt (rt0, rt1), UNPACKED. mrcal.c:5180-5506. Returns
    (err float, d_rt0[6] floats, d_rt1[6] floats, convergent)"""
    cst = lambda v: [D(c) for c in v]
    if rt0 is not None:
        r0 = [D.var(rt0[i], i)     for i in range(3)]
        t0 = [D.var(rt0[3+i], 3+i) for i in range(3)]
        t_r0   = [-c for c in rotate(r0, t0, True)]
        v0_ref = rotate(r0, cst(v0), True)
    else:
        v0_ref, t_r0 = cst(v0), cst((0, 0, 0))
    if rt1 is not None:
        r1 = [D.var(rt1[i], 6+i) for i in range(3)]
        t1 = [D.var(rt1[3+i], 9+i) for i in range(3)]
        v0_cam1 = rotate(r1, v0_ref, False)
        if rt0 is not None:
            t_10 = rotate(r1, t_r0, False)
            t_10 = [t_10[i] + t1[i] for i in range(3)]
        else:
            t_10 = t1
    else:
        v0_cam1, t_10 = v0_ref, t_r0
    err, conv = tri_error(cst(v1), v0_cam1, t_10)
    return err.x, err.g[:6], err.g[6:], conv


def noise_envelope_x(x, K):
    """|x_double - x_exact| allowed by the cancellation in 2 - 2 cos th: the
    cosine carries K roundings, dx/d(th_sq) = 2/x"""
    x = np.abs(np.asarray(x, dtype=float))
    return 1e-6*x + K*EPS/np.maximum(x, 1e-300)


def noise_envelope_J_rel(x, K):
    """|J_double - J_exact| / max|J_row| allowed: d(th_sq)/th_sq = K eps/(x/2)^2 ..."""
    x = np.abs(np.asarray(x, dtype=float))
    return 1e-6 + K*EPS/np.maximum(x*x, 1e-300)
