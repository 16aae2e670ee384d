"""The lens-model math the kernels compile (mrcal_amd/csrc/lens_models.hpp,
device_math.hpp), built for the HOST as a test-only library
(tests/hostcheck), against

  - the known answers of the reference's test/test-projections.py:336-432
    (literal pixel values; that test's own bar is 1e-2 RMS, the literals carry
    ~9 digits)
  - the reference's own mrcal_project() (oracle/_ref) on seeded random points:
    q, dq/dp and dq/dintrinsics within 1e-6 relative (in practice ~1e-12)
  - the reference's poseutils for the rotation composition and its gradients

No GPU needed: this is what pins the projection math in the CPU suite."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

from conftest import ROOT, relative_error
from mrcal_amd._cabi import Lensmodel

HERE = os.path.join(ROOT, "tests", "hostcheck")
PROJ = dict(OPENCV=0, STEREOGRAPHIC=1, LONLAT=2, LATLON=3, CAHVOR=4, CAHVORE=5)
REL_TOL = 1e-6


@pytest.fixture(scope="module")
def hostlib():
    so, src = os.path.join(HERE, "libhostcheck.so"), os.path.join(HERE, "hostcheck.cpp")
    deps = [src] + [os.path.join(ROOT, "mrcal_amd", "csrc", f) for f in ("lens_models.hpp", "device_math.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared",
                               "-Wno-unused-function", "-o", so, src])
    L = C.CDLL(so)
    vp = C.c_void_p
    L.hostcheck_project.restype  = C.c_int
    L.hostcheck_project.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_double]
    L.hostcheck_project_splined.restype  = None
    L.hostcheck_project_splined.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_double]
    L.hostcheck_compose_rt.restype  = None
    L.hostcheck_compose_rt.argtypes = [vp]*7
    L.hostcheck_R_from_r.restype  = None
    L.hostcheck_R_from_r.argtypes = [vp]*3
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def model_key(lensmodel):
    """(PROJ id, ndist, linearity) of a parametric model name"""
    if lensmodel.startswith("LENSMODEL_CAHVORE"):
        return PROJ["CAHVORE"], 8, float(lensmodel.split("linearity=")[1])
    table = dict(LENSMODEL_PINHOLE=("OPENCV",0), LENSMODEL_OPENCV4=("OPENCV",4), LENSMODEL_OPENCV5=("OPENCV",5),
                 LENSMODEL_OPENCV8=("OPENCV",8), LENSMODEL_OPENCV12=("OPENCV",12),
                 LENSMODEL_STEREOGRAPHIC=("STEREOGRAPHIC",0), LENSMODEL_LONLAT=("LONLAT",0),
                 LENSMODEL_LATLON=("LATLON",0), LENSMODEL_CAHVOR=("CAHVOR",5))
    k, n = table[lensmodel]
    return PROJ[k], n, 0.0


def host_project(L, lensmodel, intrinsics, p):
    proj, ndist, lin = model_key(lensmodel)
    p = np.ascontiguousarray(p, dtype=float)
    N = p.shape[0]
    q, dq_dp, dq_dk = np.zeros((N,2)), np.zeros((N,2,3)), np.zeros((N,2,max(ndist,1)))
    intr = np.ascontiguousarray(intrinsics, dtype=float)
    nfail = L.hostcheck_project(proj, ndist, _ptr(q), _ptr(dq_dp), _ptr(dq_dk), _ptr(p), N, _ptr(intr), lin)
    assert 0 <= nfail < 1000, "unknown model"
    return q, dq_dp, dq_dk[:,:,:ndist], nfail


def ref_project(ref_api, lensmodel, intrinsics, p):
    """the reference's mrcal_project() with both gradients"""
    clib = ref_api.clib
    clib.mrcal_project.restype  = C.c_bool
    clib.mrcal_project.argtypes = [C.c_void_p]*4 + [C.c_int, C.POINTER(Lensmodel), C.c_void_p]
    m = Lensmodel()
    assert clib.mrcal_lensmodel_from_name(C.byref(m), lensmodel.encode())
    p = np.ascontiguousarray(p, dtype=float)
    N, Ni = p.shape[0], len(intrinsics)
    q, dq_dp, dq_di = np.zeros((N,2)), np.zeros((N,2,3)), np.zeros((N,2,Ni))
    intr = np.ascontiguousarray(intrinsics, dtype=float)
    assert clib.mrcal_project(_ptr(q), _ptr(dq_dp), _ptr(dq_di), _ptr(p), N, C.byref(m), _ptr(intr))
    return q, dq_dp, dq_di


P3 = np.array(((1.0, 2.0, 10.0), (-1.1, 0.3, 1.0), (-0.9, -1.5, 1.0)))
CAHVORE_INTR = (4842.918, 4842.771, 1970.528, 1085.302, -0.001, 0.002, -0.637, -0.002, 0.016)

# reference: test/test-projections.py:341-432. (model, intrinsics, q)
KNOWN = [
    ("LENSMODEL_PINHOLE", (1512., 1112, 500., 333.), [[651.2, 555.4]]),
    ("LENSMODEL_OPENCV4", (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002),
     [[651.27371, 555.23042], [-1223.38516, 678.01468], [-1246.7310448, -1822.799928]]),
    ("LENSMODEL_OPENCV5", (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019),
     [[651.2740691, 555.2309482], [-1292.8121176, 691.9401448], [-1987.550162, -2730.85863427]]),
    ("LENSMODEL_OPENCV8", (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019, 0.014, -0.056, 0.050),
     [[651.1885442, 555.10514968], [-1234.45480366, 680.23499814], [-770.03274263, -1238.4871943]]),
    ("LENSMODEL_CAHVOR", CAHVORE_INTR,
     [[2143.17840406, 1442.93419919], [-92.63813066, 1653.09646897], [-249.83199315, -2606.46477164]]),
    ("LENSMODEL_CAHVORE_linearity=0.00", CAHVORE_INTR + (1e-8, 2e-8, 3e-8),
     [[2140.340769278752759, 1437.371480086463635], [496.634661939782575, 1493.316705796434917],
      [970.117888562484495, -568.301135889864668]]),
    ("LENSMODEL_CAHVORE_linearity=0.00", CAHVORE_INTR + (1e-2, 2e-2, 3e-2),
     [[2140.342263050081783, 1437.374408380910836], [491.682975341940221, 1494.659342555658441],
      [962.730552352575160, -580.643118338666000]]),
    ("LENSMODEL_CAHVORE_linearity=0.40", CAHVORE_INTR + (1e-2, 2e-2, 3e-2),
     [[2140.788976358770469, 1438.250116781426641], [426.278593220184689, 1512.393568241352796],
      [882.926242407330619, -713.971745152981612]]),
]


@pytest.mark.parametrize("lensmodel,intrinsics,q_known", KNOWN, ids=[k[0]+"#"+str(i) for i,k in enumerate(KNOWN)])
def test_known_answers(hostlib, lensmodel, intrinsics, q_known):
    q_known = np.array(q_known)
    q, _, _, nfail = host_project(hostlib, lensmodel, intrinsics, P3[:len(q_known)])
    assert nfail == 0
    # the literals carry 7-9 significant digits
    assert np.abs(q - q_known).max() < 2e-6*np.abs(q_known).max()


def test_known_answers_per_point_intrinsics(hostlib):
    """STEREOGRAPHIC, LATLON, LONLAT: the reference test gives each point its own intrinsics"""
    intr = np.array(((1512., 1112, 500., 333.), (1502., 1112, 500., 433.), (1522., 1112, 500., 533.)))
    known = dict(
        LENSMODEL_STEREOGRAPHIC = [[649.35582325, 552.6874014], [-813.05440267, 698.1222302], [-408.67354332, -573.48815174]],
        LENSMODEL_LATLON        = [[647.79131656, 552.50386255], [-718.86844854, 757.09995546], [-204.73403533, -559.86662025]],
        LENSMODEL_LONLAT        = [[650.69900257, 551.44238248], [-751.13786254, 654.42977413], [-615.34458492, -400.73749463]])
    for model, qk in known.items():
        for i in range(3):
            q, _, _, _ = host_project(hostlib, model, intr[i], P3[i:i+1])
            assert np.abs(q[0] - np.array(qk[i])).max() < 2e-6*np.abs(qk[i]).max(), (model, i)


RANDOM_MODELS = [
    ("LENSMODEL_PINHOLE",       (1512., 1112, 500., 333.)),
    ("LENSMODEL_STEREOGRAPHIC", (1512., 1112, 500., 333.)),
    ("LENSMODEL_LONLAT",        (1200., 1150, 500., 333.)),
    ("LENSMODEL_LATLON",        (1200., 1150, 500., 333.)),
    ("LENSMODEL_OPENCV4",  (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002)),
    ("LENSMODEL_OPENCV5",  (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019)),
    ("LENSMODEL_OPENCV8",  (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019, 0.014, -0.056, 0.050)),
    ("LENSMODEL_OPENCV12", (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019, 0.014, -0.056, 0.050,
                            0.003, -0.002, 0.001, 0.004)),
    ("LENSMODEL_CAHVOR",   CAHVORE_INTR),
    ("LENSMODEL_CAHVORE_linearity=0.00",  CAHVORE_INTR + (1e-2, 2e-2, 3e-2)),
    ("LENSMODEL_CAHVORE_linearity=0.40",  CAHVORE_INTR + (1e-2, 2e-2, 3e-2)),
    ("LENSMODEL_CAHVORE_linearity=-0.30", CAHVORE_INTR + (3e-3, -1e-2, 2e-2)),
]


@pytest.mark.parametrize("lensmodel,intrinsics", RANDOM_MODELS, ids=[m[0] for m in RANDOM_MODELS])
def test_against_reference_project(hostlib, ref_api, lensmodel, intrinsics):
    rng = np.random.RandomState(3)
    N = 200
    p = np.column_stack((rng.uniform(-1.2, 1.2, N), rng.uniform(-1.0, 1.0, N), rng.uniform(0.8, 6.0, N)))
    q, dq_dp, dq_dk, nfail = host_project(hostlib, lensmodel, intrinsics, p)
    assert nfail == 0
    qr, dq_dp_r, dq_di_r = ref_project(ref_api, lensmodel, intrinsics, p)
    assert relative_error(q, qr).max() < REL_TOL
    assert relative_error(dq_dp, dq_dp_r).max() < REL_TOL
    if dq_dk.shape[2] > 0:
        assert relative_error(dq_dk, dq_di_r[:,:,4:]).max() < REL_TOL
    # the core columns are what the kernels form themselves: (q-c)/f and 1
    f, c = np.array(intrinsics[:2]), np.array(intrinsics[2:4])
    assert relative_error((q - c)/f, np.stack((dq_di_r[:,0,0], dq_di_r[:,1,1]), axis=1)).max() < REL_TOL


SPLINED = [("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=200", 3, 11, 8, 200.),
           ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=2_Nx=11_Ny=8_fov_x_deg=200", 2, 11, 8, 200.),
           ("LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=30_Ny=20_fov_x_deg=150", 3, 30, 20, 150.)]


@pytest.mark.parametrize("lensmodel,order,Nx,Ny,fov", SPLINED, ids=[s[0][31:] for s in SPLINED])
def test_splined_against_reference(hostlib, ref_api, lensmodel, order, Nx, Ny, fov):
    rng = np.random.RandomState(5)
    intr = np.concatenate(((1500., 1800., 1499.5, 999.5), rng.uniform(-0.05, 0.05, 2*Nx*Ny)))
    N = 300
    # wide field, and a few points beyond the knots (the clamped segments)
    p = np.column_stack((rng.uniform(-3, 3, N), rng.uniform(-2, 2, N), rng.uniform(0.3, 3.0, N)))
    p[:5] *= np.array((8., 8., 1.))
    q, dq_dp, dq_dfxy = np.zeros((N,2)), np.zeros((N,2,3)), np.zeros((N,2))
    ivar0, coef = np.zeros(N, dtype=np.int32), np.zeros((N,8))
    pc = np.ascontiguousarray(p)
    hostlib.hostcheck_project_splined(_ptr(q), _ptr(dq_dp), _ptr(dq_dfxy), _ptr(ivar0), _ptr(coef),
                                      _ptr(pc), N, _ptr(intr), order, Nx, Ny, fov)
    qr, dq_dp_r, dq_di_r = ref_project(ref_api, lensmodel, intr, p)
    assert relative_error(q, qr).max() < REL_TOL
    assert relative_error(dq_dp, dq_dp_r).max() < REL_TOL
    # densify our sparse intrinsics gradient like the reference does (mrcal.c:2965-2991)
    n = order + 1
    dense = np.zeros_like(dq_di_r)
    dense[:,0,0] = dq_dfxy[:,0]; dense[:,1,1] = dq_dfxy[:,1]
    dense[:,0,2] = 1.;           dense[:,1,3] = 1.
    for i in range(N):
        for jy in range(n):
            for jx in range(n):
                k = ivar0[i] + jy*2*Nx + jx*2
                v = coef[i,jx]*coef[i,4+jy]
                dense[i,0,k]   = v*intr[0]
                dense[i,1,k+1] = v*intr[1]
    assert relative_error(dense, dq_di_r).max() < REL_TOL


@pytest.mark.parametrize("order,Nx,Ny,fov", ((3, 30, 20, 150.), (2, 16, 12, 120.), (3, 11, 8, 100.)))
def test_splined_projection_by_rows_is_the_projection_by_corners(hostlib, order, Nx, Ny, fov):
    """Round 6: board_splined_rows_kernel gives every Jacobian ROW a lane, which projects ONE image coordinate
    (lens_models.hpp project_splined_row: the shared part + one surface). It is project_splined()'s own arithmetic for
    that coordinate, in its order: the same BITS in q, dq/dp, dq/df, the basis values and the first control point, for
    either coordinate, inside the grid and in the clamped segments beyond it"""
    rng = np.random.RandomState(6)
    intr = np.concatenate(((1500., 1800., 1499.5, 999.5), rng.uniform(-0.05, 0.05, 2*Nx*Ny)))
    N = 500
    p = np.column_stack((rng.uniform(-3, 3, N), rng.uniform(-2, 2, N), rng.uniform(0.3, 3.0, N)))
    p[:8] *= np.array((8., 8., 1.))
    pc = np.ascontiguousarray(p)
    def arrays(): return np.zeros((N,2)), np.zeros((N,2,3)), np.zeros((N,2)), np.zeros(N, dtype=np.int32), np.zeros((N,8))
    q, g, df, iv, coef = arrays()
    hostlib.hostcheck_project_splined(_ptr(q), _ptr(g), _ptr(df), _ptr(iv), _ptr(coef), _ptr(pc), N, _ptr(intr), order, Nx, Ny, fov)
    q2, g2, df2, iv2, coef2 = arrays()
    coef_k1 = np.zeros((N,8))
    f = hostlib.hostcheck_project_splined_rows
    f.restype, f.argtypes = None, [C.c_void_p]*7 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
    f(_ptr(q2), _ptr(g2), _ptr(df2), _ptr(iv2), _ptr(coef2), _ptr(coef_k1), _ptr(pc), N, _ptr(intr), order, Nx, Ny, fov)
    assert np.array_equal(iv, iv2)
    assert np.array_equal(q, q2) and np.array_equal(g, g2) and np.array_equal(df, df2)
    assert np.array_equal(coef, coef2) and np.array_equal(coef, coef_k1)


def test_compose_rt_and_R_from_r(hostlib, ref_api):
    """rotation composition with all its gradients, incl. the tiny-angle and
    near-pi branches, against the reference's poseutils"""
    clib = ref_api.clib
    vp, ci = C.c_void_p, C.c_int
    clib.mrcal_compose_rt_full.restype  = None
    clib.mrcal_compose_rt_full.argtypes = [vp, ci] + [vp, ci, ci]*6 + [vp, ci, vp, ci, C.c_bool, C.c_bool]
    clib.mrcal_R_from_r_full.restype  = None
    clib.mrcal_R_from_r_full.argtypes = [vp, ci, ci, vp, ci, ci, ci, vp, ci]
    rng = np.random.RandomState(9)
    cases = [rng.uniform(-1.5, 1.5, 12) for _ in range(20)]
    cases += [np.r_[np.zeros(3), rng.uniform(-1,1,3), rng.uniform(-1,1,6)],            # r0 = 0
              np.r_[rng.uniform(-1,1,3), rng.uniform(-1,1,3), np.zeros(3), 1., 2., 3.], # r1 = 0
              np.r_[1e-9, 0, 1e-10, 0.1, 0.2, 0.3, 0.5, -0.4, 0.3, 1., 2., 3.],
              np.r_[0, 0, 3.0, 0.1, 0.2, 0.3, 0, 0, 3.2, 1., 2., 3.],
              np.r_[0, 0, 3.0, 0.1, 0.2, 0.3, 0, 0, 2*np.pi-3.0-1e-9, 1., 2., 3.]]
    for c in cases:
        rt0, rt1 = np.ascontiguousarray(c[:6]), np.ascontiguousarray(c[6:])
        out, g = np.zeros(6), [np.zeros((3,3)) for _ in range(4)]
        hostlib.hostcheck_compose_rt(_ptr(out), *[_ptr(a) for a in g], _ptr(rt0), _ptr(rt1))
        ro, rg = np.zeros(6), [np.zeros((3,3)) for _ in range(4)]
        # (rt_out, dr_dr0, dr_dr1, dt_dr0, dt_dr1=NULL, dt_dt0=NULL, dt_dt1); strides 0 = contiguous
        clib.mrcal_compose_rt_full(_ptr(ro), 0,
                                   _ptr(rg[0]), 0, 0,  _ptr(rg[1]), 0, 0,
                                   _ptr(rg[2]), 0, 0,  None, 0, 0,  None, 0, 0,
                                   _ptr(rg[3]), 0, 0,
                                   _ptr(rt0), 0, _ptr(rt1), 0, False, False)
        assert relative_error(out, ro).max() < REL_TOL, c
        for a, b in zip(g, rg):
            assert relative_error(a, b).max() < REL_TOL, c
    for r in [rng.uniform(-2,2,3) for _ in range(10)] + [np.zeros(3), np.array((1e-20,0,0))]:
        r = np.ascontiguousarray(r)
        R, dR = np.zeros((3,3)), np.zeros((3,3,3))
        hostlib.hostcheck_R_from_r(_ptr(R), _ptr(dR), _ptr(r))
        Rr, dRr = np.zeros((3,3)), np.zeros((3,3,3))
        clib.mrcal_R_from_r_full(_ptr(Rr), 0, 0, _ptr(dRr), 0, 0, 0, _ptr(r), 0)
        assert relative_error(R, Rr).max() < REL_TOL
        assert relative_error(dR, dRr).max() < REL_TOL
