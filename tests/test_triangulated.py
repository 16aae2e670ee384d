"""Triangulated-point observations: the SfM-style residual, one
row per pair of observations of a point.

CPU: mrcal_unproject() (host code of libmrcal_amd.so) against the reference's;
     the pair residual + its 12 derivatives (host build of
     mrcal_amd/csrc/triangulation.hpp) against the reference's callback.
GPU: optimizer_callback() x, J, CSR structure against the reference; the solve.

Scene: the shape of the reference's test/test-sfm-triangulated-points.py cut
down: PINHOLE f=600 cameras (one at the reference), points on a plane, pixel
noise, intrinsics locked, extrinsics optimized, unity_cam01 regularization."""
import ctypes as C
import numpy as np
import pytest

from conftest import relative_error, ROOT
from mrcal_amd._cabi import Lensmodel
from mrcal_amd.synthetic import copy_inputs

REL_TOL = 1e-6


def R_from_r(r):
    th = np.linalg.norm(r)
    if th < 1e-12: return np.eye(3)
    k = r/th
    K = np.array(((0,-k[2],k[1]),(k[2],0,-k[0]),(-k[1],k[0],0)))
    return np.eye(3) + np.sin(th)*K + (1-np.cos(th))*(K@K)


# (the generator lives with the other synthetic workloads since round 5: bench.py times these configurations too)
from mrcal_amd.synthetic import make_sfm_problem as sfm_problem


# ------------------------------------------------------------------ CPU ---
UNPROJECT_MODELS = [
    ("LENSMODEL_PINHOLE",       (1512., 1112, 500., 333.)),
    ("LENSMODEL_STEREOGRAPHIC", (1512., 1112, 500., 333.)),
    ("LENSMODEL_LONLAT",        (1200., 1150, 500., 333.)),
    ("LENSMODEL_LATLON",        (1200., 1150, 500., 333.)),
    ("LENSMODEL_OPENCV4",  (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002)),
    ("LENSMODEL_OPENCV8",  (1512., 1112, 500., 333., -0.012, 0.035, -0.001, 0.002, 0.019, 0.014, -0.056, 0.050)),
    ("LENSMODEL_CAHVOR",   (4842.918, 4842.771, 1970.528, 1085.302, -0.001, 0.002, -0.637, -0.002, 0.016)),
    ("LENSMODEL_CAHVORE_linearity=0.40", (4842.918, 4842.771, 1970.528, 1085.302, -0.001, 0.002, -0.637, -0.002, 0.016, 0., 0., 0.)),
]


@pytest.mark.parametrize("lensmodel,intrinsics", UNPROJECT_MODELS, ids=[m[0] for m in UNPROJECT_MODELS])
def test_unproject_matches_reference(amd, ref_api, lensmodel, intrinsics):
    """host code: runs without a GPU"""
    rng = np.random.RandomState(2)
    intr = np.array(intrinsics, dtype=float)
    q = np.ascontiguousarray(intr[2:4] + rng.uniform(-0.6, 0.6, (200,2))*intr[:2])
    out = []
    for lib in (amd._lib.lib, ref_api.clib):
        lib.mrcal_unproject.restype  = C.c_bool
        lib.mrcal_unproject.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Lensmodel), C.c_void_p]
        m = Lensmodel()
        assert lib.mrcal_lensmodel_from_name(C.byref(m), lensmodel.encode())
        v = np.zeros((q.shape[0],3))
        assert lib.mrcal_unproject(v.ctypes.data, q.ctypes.data, q.shape[0], C.byref(m), intr.ctypes.data)
        out.append(v / np.linalg.norm(v, axis=1, keepdims=True))   # "may have any length"
    assert np.isfinite(out[0]).all()
    assert np.abs(out[0] - out[1]).max() < 1e-8


def test_unproject_of_many_points_is_the_unprojection_of_each(amd):
    """round 6: from 2048 points on mrcal_unproject() deals ranges of the points to threads (the 57 000 observations of
    BASELINE configuration 4 were 10 ms of one core in front of a 12 ms solve). A point's vector does not depend on its
    company: the same bits as 200 points at a time, which one thread takes. Host code: runs without a GPU"""
    rng = np.random.RandomState(3)
    intr = np.array(UNPROJECT_MODELS[5][1], dtype=float)
    q = np.ascontiguousarray(intr[2:4] + rng.uniform(-0.6, 0.6, (20001,2))*intr[:2])
    lib = amd._lib.lib
    lib.mrcal_unproject.restype  = C.c_bool
    lib.mrcal_unproject.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Lensmodel), C.c_void_p]
    m = Lensmodel()
    assert lib.mrcal_lensmodel_from_name(C.byref(m), UNPROJECT_MODELS[5][0].encode())
    v = np.zeros((q.shape[0],3))
    assert lib.mrcal_unproject(v.ctypes.data, q.ctypes.data, q.shape[0], C.byref(m), intr.ctypes.data)
    w = np.zeros_like(v)
    for i0 in range(0, q.shape[0], 200):
        n = min(200, q.shape[0] - i0)
        assert lib.mrcal_unproject(w[i0:].ctypes.data, q[i0:].ctypes.data, n, C.byref(m), intr.ctypes.data)
    assert np.isfinite(v).all() and np.array_equal(v, w)


SPLINED_MODEL = "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120"


def _unproject_cases():
    cases = list(UNPROJECT_MODELS)
    rng = np.random.RandomState(12)
    cases.append((SPLINED_MODEL, tuple(np.r_[1100., 1100., 500., 333., rng.uniform(-0.02, 0.02, 2*11*8)])))
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("lensmodel,intrinsics", _unproject_cases(), ids=[m[0][:30] for m in _unproject_cases()])
def test_python_unproject(amd_api, ref_api, lensmodel, intrinsics):
    """mrcal_amd.unproject() (GPU): shapes, normalize, against the reference's C
    function, project(unproject(q)) == q, and the gradients
    (mrcal/projections.py:112-395) against central differences and against the
    identities they are derived from"""
    rng  = np.random.RandomState(4)
    intr = np.array(intrinsics, dtype=float)
    Ni   = len(intr)
    q    = intr[2:4] + rng.uniform(-400, 400, size=(5, 7, 2))
    v    = amd_api.unproject(q, lensmodel, intr)
    assert v.shape == (5, 7, 3)
    vn = amd_api.unproject(q, lensmodel, intr, normalize=True)
    assert np.allclose(np.linalg.norm(vn, axis=-1), 1.0)
    assert np.allclose(np.cross(vn, v), 0, atol=1e-9)
    # the reference's mrcal_unproject(): the same direction
    vr = ref_api.unproject(q, lensmodel, intr, normalize=True)
    assert np.abs(vn - vr).max() < 1e-8
    # the reference's mrcal_project() brings them back to the pixels
    m = ref_api.lib.lensmodel(lensmodel)
    f = ref_api.clib.mrcal_project
    f.restype  = C.c_bool
    f.argtypes = [C.c_void_p]*4 + [C.c_int, C.c_void_p, C.c_void_p]
    q2 = np.empty((35, 2))
    vv = np.ascontiguousarray(v.reshape(-1, 3))
    assert f(q2.ctypes.data, None, None, vv.ctypes.data, 35, C.byref(m), intr.ctypes.data)
    assert np.abs(q2 - q.reshape(-1, 2)).max() < 1e-6

    # ---- gradients
    for normalize in (False, True):
        vg, dv_dq, dv_di = amd_api.unproject(q, lensmodel, intr, normalize=normalize, get_gradients=True)
        assert vg.shape == (5,7,3) and dv_dq.shape == (5,7,3,2) and dv_di.shape == (5,7,3,Ni)
        # the same direction as without gradients; with normalize the same vector
        assert np.allclose(np.cross(vg, vn), 0, atol=1e-9)
        if normalize:
            assert np.abs(vg - vn).max() < 1e-12
            # gradients of a unit vector are orthogonal to it
            assert np.abs(np.einsum("...k,...kj->...j", vg, dv_dq)).max() < 1e-9*np.abs(dv_dq).max()
        # project(v(q,i), i) == q identically:  dq/dv dv/dq = I,  dq/dv dv/di + dq/di = 0
        _, dq_dv, dq_di = amd_api.project(vg, lensmodel, intr, get_gradients=True)
        assert np.abs(np.einsum("...ak,...kb->...ab", dq_dv, dv_dq) - np.eye(2)).max() < 1e-8
        # (CAHVORE's E moves the entrance pupil: dq/dE depends on the LENGTH of v, so that part of the
        #  identity holds at the vector the gradients were taken at - the un-normalized one - only)
        ncheck = Ni - 3 if (lensmodel.startswith("LENSMODEL_CAHVORE") and normalize) else Ni
        assert np.abs((np.einsum("...ak,...kj->...aj", dq_dv, dv_di) + dq_di)[...,:ncheck]).max() < \
            1e-7*max(1.0, np.abs(dq_di).max())
        # central differences of the routine itself (normalized: the length convention does not enter)
        if normalize:
            dq = 1e-3
            for b in range(2):
                e = np.zeros(2); e[b] = dq
                fd = (amd_api.unproject(q + e, lensmodel, intr, normalize=True) -
                      amd_api.unproject(q - e, lensmodel, intr, normalize=True))/(2*dq)
                assert np.abs(fd - dv_dq[...,b]).max() < 1e-6*max(1e-3, np.abs(dv_dq).max())
            # (CAHVORE: unproject() refuses E != 0, so E cannot be perturbed)
            jlast = Ni-4 if lensmodel.startswith("LENSMODEL_CAHVORE") else Ni-1
            for j in list(range(min(Ni, 6))) + ([jlast] if Ni > 6 else []):
                di = 1e-6*max(1.0, abs(intr[j]))
                e = np.zeros(Ni); e[j] = di
                fd = (amd_api.unproject(q, lensmodel, intr + e, normalize=True) -
                      amd_api.unproject(q, lensmodel, intr - e, normalize=True))/(2*di)
                assert np.abs(fd - dv_di[...,j]).max() < 1e-5*max(1e-6, np.abs(dv_di[...,j]).max()) + 1e-9, j


@pytest.mark.gpu
def test_project_unproject_broadcast_over_models(amd_api):
    """mrcal.project()/unproject() broadcast over the intrinsics too (numpysane
    prototypes ((3,),(Nintrinsics,)) / ((2,),(Nintrinsics,)))"""
    rng = np.random.RandomState(5)
    lensmodel, base = UNPROJECT_MODELS[4][0], np.array(UNPROJECT_MODELS[4][1])
    intr = base + rng.uniform(-1, 1, size=(3, 1, len(base)))*np.r_[5., 5., 5., 5., 1e-3*np.ones(len(base)-4)]
    v = rng.uniform(-0.3, 0.3, size=(4, 3)); v[:,2] = 1.0
    q = amd_api.project(v, lensmodel, intr)
    assert q.shape == (3, 4, 2)
    for i in range(3):
        assert np.array_equal(q[i], amd_api.project(v, lensmodel, intr[i,0]))
    vb = amd_api.unproject(q, lensmodel, intr)
    assert vb.shape == (3, 4, 3)
    assert np.abs(vb/vb[...,2:] - v).max() < 1e-9
    vg, dv_dq, dv_di = amd_api.unproject(q, lensmodel, intr, get_gradients=True, normalize=True)
    assert dv_dq.shape == (3, 4, 3, 2) and dv_di.shape == (3, 4, 3, len(base))
    g1 = amd_api.unproject(q[1], lensmodel, intr[1,0], get_gradients=True, normalize=True)
    assert np.array_equal(vg[1], g1[0]) and np.array_equal(dv_dq[1], g1[1]) and np.array_equal(dv_di[1], g1[2])


def test_pair_residual_matches_reference_cpu(ref_api):
    """the pair residual and its derivatives (host build of triangulation.hpp)
    against the rows the reference's callback produces. No GPU"""
    import subprocess, os
    from test_lens_models_host import HERE
    so = os.path.join(HERE, "libhostcheck.so")
    src = os.path.join(HERE, "hostcheck.cpp")
    deps = [src] + [os.path.join(ROOT, "mrcal_amd", "csrc", f) for f in ("lens_models.hpp", "device_math.hpp", "triangulation.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-o", so, src])
    L = C.CDLL(so)
    L.hostcheck_tri_pair_error.restype  = C.c_double
    L.hostcheck_tri_pair_error.argtypes = [C.c_void_p]*3 + [C.c_void_p]*4

    oi, _ = sfm_problem(seed=3, noise=2.0)
    oi["do_apply_regularization"] = False
    oi["do_apply_regularization_unity_cam01"] = False
    b, x, J, _ = ref_api.optimizer_callback(no_factorization=True, **oi)
    Jd = J.toarray()
    # the C records the reference saw: observation vectors and set structure
    p = ref_api._ingest(dict(oi), callback=True)
    px, flags = p.c_tri["px"], p.c_tri["flags"]
    ice = p.c_tri["icam_extrinsics"]
    rt = oi["rt_cam_ref"]
    SR, ST = 0.1*np.pi/180., 1.0
    irow = 0
    N = len(px)
    Nsteep = 0
    for i0 in range(N):
        if flags[i0] & 1: continue
        for i1 in range(i0+1, N):
            e0, e1 = ice[i0], ice[i1]
            d0, d1, conv = np.zeros(6), np.zeros(6), C.c_int(0)
            rt0 = np.ascontiguousarray(rt[e0]) if e0 >= 0 else None
            rt1 = np.ascontiguousarray(rt[e1]) if e1 >= 0 else None
            err = L.hostcheck_tri_pair_error(d0.ctypes.data, d1.ctypes.data, C.byref(conv),
                                             np.ascontiguousarray(px[i0]).ctypes.data, np.ascontiguousarray(px[i1]).ctypes.data,
                                             rt0.ctypes.data if rt0 is not None else None,
                                             rt1.ctypes.data if rt1 is not None else None)
            assert err != -12345.0
            outlier = bool((flags[i0] | flags[i1]) & 2)
            if outlier:
                assert x[irow] == 0 and not Jd[irow].any()
            else:
                assert relative_error(err, x[irow]) < REL_TOL
                scale = np.array((SR,SR,SR,ST,ST,ST))
                if e0 >= 0: assert relative_error(d0*scale, Jd[irow, 6*e0:6*e0+6]).max() < REL_TOL
                if e1 >= 0: assert relative_error(d1*scale, Jd[irow, 6*e1:6*e1+6]).max() < REL_TOL
                Nsteep += 1
            irow += 1
            if flags[i1] & 1: break
    assert irow == x.size and Nsteep > 50


def enumerate_pairs(flags):
    """(i0,i1) of every measurement row, in row order: mrcal.c:5180-5290 (a
    point's observations are consecutive; bit 0 of the flags = last of its set)"""
    pairs = []
    N = len(flags)
    for i0 in range(N):
        if flags[i0] & 1: continue
        for i1 in range(i0+1, N):
            pairs.append((i0, i1))
            if flags[i1] & 1: break
    return pairs


def exact_rows(rows, pairs, px, flags, ice, rt):
    """the mpmath evaluation (mp_triangulated.py) of the given rows: x, and the
    packed-state gradient wrt (camera 0 extrinsics, camera 1 extrinsics) as
    present in the row. Rows of outlier pairs are skipped. -> {row: (x, J)}"""
    import mp_triangulated as M
    scale = np.array((M.SCALE_ROTATION_CAMERA,)*3 + (M.SCALE_TRANSLATION_CAMERA,)*3)
    out = {}
    for r in rows:
        i0, i1 = pairs[r]
        if (flags[i0] | flags[i1]) & 2: continue
        e0, e1 = ice[i0], ice[i1]
        ex, g0, g1, _ = M.pair_error(px[i0], px[i1], rt[e0] if e0 >= 0 else None, rt[e1] if e1 >= 0 else None)
        Je = []
        if e0 >= 0: Je += list(np.array([float(v) for v in g0])*scale)
        if e1 >= 0: Je += list(np.array([float(v) for v in g1])*scale)
        out[int(r)] = (float(ex), np.array(Je))
    return out


K_ENVELOPE = 16.   # roundings of the cosine, with margin: observed 5 (x), 3.5 (J)


def test_pair_residual_rounding_envelope_cpu(ref_api):
    """The third opinion behind the small-angle tolerance of the full-size
    test (test_full_size.py::test_sfm_configuration_full_size): at 3000 points,
    for the 150 pairs with the smallest residual and 60 random ones, BOTH the
    reference's rows and the host build of triangulation.hpp are within
    K eps/|x| (x) and K eps/x^2 relative (J) of the 60-digit evaluation of the
    same formula. No GPU"""
    import mp_triangulated as M
    from test_lens_models_host import HERE
    import os
    L = C.CDLL(os.path.join(HERE, "libhostcheck.so"))
    L.hostcheck_tri_pair_error.restype  = C.c_double
    L.hostcheck_tri_pair_error.argtypes = [C.c_void_p]*3 + [C.c_void_p]*4
    oi, _ = sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=3000, seed=6, noise=0.3)
    oi["do_apply_regularization"] = False
    oi["do_apply_regularization_unity_cam01"] = False
    _, x, J, _ = ref_api.optimizer_callback(no_factorization=True, **oi)
    p = ref_api._ingest(dict(oi), callback=True)
    px, flags, ice = p.c_tri["px"], p.c_tri["flags"], p.c_tri["icam_extrinsics"]
    rt = oi["rt_cam_ref"]
    pairs = enumerate_pairs(flags)
    assert len(pairs) == x.size
    rows = list(np.argsort(np.abs(x))[:150]) + list(np.random.RandomState(0).choice(x.size, 60, replace=False))
    exact = exact_rows(rows, pairs, px, flags, ice, rt)
    assert len(exact) > 190
    scale = np.array((M.SCALE_ROTATION_CAMERA,)*3 + (M.SCALE_TRANSLATION_CAMERA,)*3)
    Nsmall = 0
    for r, (xe, Je) in exact.items():
        i0, i1 = pairs[r]
        e0, e1 = ice[i0], ice[i1]
        rt0 = np.ascontiguousarray(rt[e0]) if e0 >= 0 else None
        rt1 = np.ascontiguousarray(rt[e1]) if e1 >= 0 else None
        d0, d1, conv = np.zeros(6), np.zeros(6), C.c_int(0)
        xo = L.hostcheck_tri_pair_error(d0.ctypes.data, d1.ctypes.data, C.byref(conv),
                                        np.ascontiguousarray(px[i0]).ctypes.data, np.ascontiguousarray(px[i1]).ctypes.data,
                                        rt0.ctypes.data if rt0 is not None else None,
                                        rt1.ctypes.data if rt1 is not None else None)
        Jo = np.concatenate(([d0*scale] if e0 >= 0 else []) + ([d1*scale] if e1 >= 0 else []))
        Jr = J.data[J.indptr[r]:J.indptr[r+1]]
        tol_x = M.noise_envelope_x(xe, K_ENVELOPE)
        tol_J = M.noise_envelope_J_rel(xe, K_ENVELOPE)*np.abs(Je).max()
        assert abs(xo   - xe) <= tol_x and abs(x[r] - xe) <= tol_x, (r, xe, xo, x[r])
        assert np.abs(Jo - Je).max() <= tol_J and np.abs(Jr - Je).max() <= tol_J, (r, xe)
        Nsmall += abs(xe) < 1e-4
    assert Nsmall > 50


# ------------------------------------------------------------------ GPU ---
@pytest.mark.gpu
@pytest.mark.parametrize("lensmodel", ("LENSMODEL_PINHOLE", "LENSMODEL_OPENCV4"))
def test_callback_matches_reference(amd, ref_api, lensmodel):
    from test_callback_parity import compare_callbacks
    oi, _ = sfm_problem(lensmodel=lensmodel, seed=1, noise=1.0)
    for unity in (True, False):
        oi["do_apply_regularization_unity_cam01"] = unity
        compare_callbacks(amd.optimizer_callback(no_factorization=True, **oi),
                          ref_api.optimizer_callback(no_factorization=True, **oi), f"{lensmodel} unity={unity}")
    assert amd.num_measurements(**oi) == ref_api.num_measurements(**oi)
    assert amd.num_measurements_points_triangulated(**oi) == ref_api.num_measurements_points_triangulated(**oi)
    assert amd.measurement_index_regularization(**oi) == ref_api.measurement_index_regularization(**oi)


@pytest.mark.gpu
def test_divergent_rays_and_x_only(amd, ref_api):
    """a badly wrong seed makes rays diverge: the penalty branch"""
    from test_callback_parity import compare_callbacks
    oi, _ = sfm_problem(seed=4, noise=0.3)
    oi["rt_cam_ref"][1,:3] += (0.0, 0.3, 0.0)     # yaw one camera outwards
    ra = amd.optimizer_callback(no_factorization=True, **oi)
    rr = ref_api.optimizer_callback(no_factorization=True, **oi)
    compare_callbacks(ra, rr, "divergent")
    xa = amd.optimizer_callback(no_jacobian=True, no_factorization=True, **oi)[1]
    assert relative_error(xa, rr[1]).max() < REL_TOL


@pytest.mark.gpu
def test_solve_matches_checker_and_truth(amd, ref_api):
    oi, truth = sfm_problem(seed=7, noise=0.3, Npoints=80)
    oi["do_apply_outlier_rejection"] = True
    oa, orr = copy_inputs(oi), copy_inputs(oi)
    sa = amd.optimize(**oa)
    sr = ref_api.optimize(**orr)
    assert abs(sa["rms_reproj_error__pixels"] - sr["rms_reproj_error__pixels"]) < 1e-6*sr["rms_reproj_error__pixels"]
    assert sa["Noutliers_triangulated_point"] == sr["Noutliers_triangulated_point"]
    # the outlier bits in the C records of the triangulated observations (mrcal.c:4225, 4375 write them into the
    # caller's array): the same observations marked
    assert np.array_equal(amd._api._last_triangulated_flags, ref_api._last_triangulated_flags)
    assert np.abs(oa["rt_cam_ref"] - orr["rt_cam_ref"]).max() < 1e-5
    # unity_cam01 fixes the scale: camera 1 is 1m from the reference; the
    # geometry comes out close to the truth
    assert abs(np.linalg.norm(oa["rt_cam_ref"][0,3:]) - 1.0) < 1e-3
    assert np.abs(oa["rt_cam_ref"][:,:3] - truth["rt_cam_ref"][:,:3]).max() < 0.01


def compare_callbacks_with_pairs(res_amd, res_ref, m0, m1, what=""):
    """compare_callbacks() (bit-exact b_packed and CSR structure, x and J to 1e-6 by the reference's relative
    error) with the bar of the triangulated rows [m0,m1) widened by the rounding envelope of their formula and
    by nothing else: the residual is 2 sqrt(2 - 2 cos th)-like (triangulation.cc:767-805) and ANY double
    evaluation of it is off by ~K eps/|x| in x and ~K eps/x^2 relative in its gradient (mp_triangulated.py;
    test_pair_residual_rounding_envelope_cpu holds the reference's rows and ours to K = 16 against a
    60-digit evaluation). Two implementations may differ by twice that"""
    import mp_triangulated as M
    b_a, x_a, J_a, _ = res_amd
    b_r, x_r, J_r, _ = res_ref
    assert np.array_equal(b_a, b_r), f"{what}: b_packed differs"
    assert np.array_equal(J_a.indptr,  J_r.indptr),  f"{what}: CSR rowptr differs"
    assert np.array_equal(J_a.indices, J_r.indices), f"{what}: CSR colidx differs"
    tri = np.zeros(x_r.shape, dtype=bool); tri[m0:m1] = True
    if (~tri).any(): assert relative_error(x_a[~tri], x_r[~tri]).max() < REL_TOL, what
    assert np.all(np.abs(x_a - x_r)[tri] <= np.maximum(REL_TOL*np.abs(x_r[tri]), 2*M.noise_envelope_x(x_r[tri], K_ENVELOPE))), what
    row_of = np.repeat(np.arange(len(x_r)), np.diff(J_r.indptr))
    maxJ_row = np.maximum.reduceat(np.abs(J_r.data), J_r.indptr[:-1])
    tolJ = np.where(tri, np.maximum(REL_TOL, 2*M.noise_envelope_J_rel(x_r, K_ENVELOPE)), 0.0)*maxJ_row
    excess = np.abs(J_a.data - J_r.data) - np.maximum(tolJ[row_of], REL_TOL*(np.abs(J_a.data) + np.abs(J_r.data))/2. + REL_TOL*1e-6)
    assert excess.max() <= 0, f"{what}: J differs beyond the bar in row {row_of[np.argmax(excess)]}"


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("lensmodel,Npoints,Nboard_frames", (("LENSMODEL_PINHOLE", 60,    5),
                                                             ("LENSMODEL_OPENCV4", 2500,  300),
                                                             # BASELINE.json's configuration 5 at its stated size
                                                             # (callback only: the solve is compared one size down)
                                                             ("LENSMODEL_OPENCV4", 20000, 400)))
def test_boards_and_triangulated_in_one_problem(amd, ref_api, lensmodel, Npoints, Nboard_frames):
    """BASELINE.json's configuration 5 shape: board frames AND triangulated points in one problem (intrinsics
    locked, extrinsics + frames optimized: mrcal.c:6043-6051). The triangulated rows sit BEHIND the board rows
    in the measurement vector (mrcal.c:5180-5653 after :4603-4898), they touch only the extrinsics columns
    while the board rows couple extrinsics and frames: the Schur elimination of the frames and the
    extrinsics-only generic rows in one solve. Callback against the reference, then the solve against the
    checker"""
    from test_callback_parity import compare_callbacks
    oi, truth = sfm_problem(lensmodel=lensmodel, Ncam=4, Npoints=Npoints, seed=9, noise=0.3, Nboard_frames=Nboard_frames)
    Nobs_board = 4*Nboard_frames
    for unity in (True, False):
        oi["do_apply_regularization_unity_cam01"] = unity
        ra = amd.optimizer_callback(no_factorization=True, **oi)
        rr = ref_api.optimizer_callback(no_factorization=True, **oi)
        m0 = amd.measurement_index_points_triangulated(**oi)
        compare_callbacks_with_pairs(ra, rr, m0, m0 + amd.num_measurements_points_triangulated(**oi),
                                     f"boards+triangulated {lensmodel} unity={unity}")
    # the layout: board rows first, the pairs behind them, then the regularization
    assert amd.num_states(**oi) == ref_api.num_states(**oi) == 6*3 + 6*Nboard_frames
    assert amd.measurement_index_points_triangulated(**oi) == ref_api.measurement_index_points_triangulated(**oi) \
        == Nobs_board*100*2
    assert amd.num_measurements_points_triangulated(**oi) == ref_api.num_measurements_points_triangulated(**oi) >= Npoints
    assert amd.measurement_index_regularization(**oi) == ref_api.measurement_index_regularization(**oi)
    J = ra[2]
    m0 = Nobs_board*200
    # a pair's row holds extrinsics columns only; a board row of a non-reference camera holds both
    assert J.indices[J.indptr[m0]:J.indptr[m0 + amd.num_measurements_points_triangulated(**oi)]].max() < 18
    assert J.indices[J.indptr[200]:J.indptr[201]].max() >= 18
    if Npoints >= 20000:
        assert amd.num_measurements_points_triangulated(**oi) > 60000
        return

    # the solve, outlier rejection on (boards AND divergent/k-sigma pairs)
    oi["do_apply_regularization_unity_cam01"] = True
    oi["do_apply_outlier_rejection"] = True
    oa, orr = copy_inputs(oi), copy_inputs(oi)
    sa = amd.optimize(**oa)
    sr = ref_api.optimize(**orr)
    assert sa["Noutliers_board"] == sr["Noutliers_board"]
    assert sa["Noutliers_triangulated_point"] == sr["Noutliers_triangulated_point"]
    from mrcal_amd._cabi import TRIANGULATED_OUTLIER
    assert np.array_equal(amd._api._last_triangulated_flags, ref_api._last_triangulated_flags)
    assert (sr["Noutliers_triangulated_point"] == 0) == (not np.any(ref_api._last_triangulated_flags & TRIANGULATED_OUTLIER))
    assert np.array_equal(oa["observations_board"][...,2] < 0, orr["observations_board"][...,2] < 0)
    assert abs(sa["rms_reproj_error__pixels"] - sr["rms_reproj_error__pixels"]) < 1e-6*sr["rms_reproj_error__pixels"]
    assert np.abs(sa["b_packed"] - sr["b_packed"]).max() < 2e-5
    assert np.abs(oa["rt_cam_ref"] - orr["rt_cam_ref"]).max() < 1e-5
    assert np.abs(oa["rt_ref_frame"] - orr["rt_ref_frame"]).max() < 1e-4
    # and the geometry is the truth's up to the noise (the boards fix the scale)
    assert np.abs(oa["rt_cam_ref"][:,:3] - truth["rt_cam_ref"][:,:3]).max() < 5e-3
    assert np.abs(oa["rt_cam_ref"][:,3:] - truth["rt_cam_ref"][:,3:]).max() < (5e-2 if Nboard_frames < 50 else 5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("jacobian_stream", (True, False))
def test_the_solvers_one_launch_writes_the_evaluations_bits(amd, jacobian_stream):
    """Round 6: inside a solve the board observations and the triangulated pairs of a problem that has both are ONE
    launch (kernels.hip board_tri_kernel: the pairs' workgroups in front of the observations'), with the Grams; a
    host-driven evaluation - optimizer_callback()'s - is board_kernel without the Grams and triangulated_kernel behind
    it. The same device functions, -ffp-contract=on: x and the CSR values the solver's launches left at its current
    point are, to the last bit, what an evaluation at that state gives - with the solve's Jacobian stream on, and off
    (the pairs' rows are streamed either way; the boards' are made on demand)"""
    from mrcal_amd.resident import Problem
    oi, _ = sfm_problem(lensmodel="LENSMODEL_OPENCV4", Ncam=4, Npoints=2500, seed=9, noise=0.3, Nboard_frames=50)
    with Problem(**copy_inputs(oi)) as p:
        p.set_jacobian_stream(jacobian_stream)
        n, _ = p.run_steps(4)
        assert n == 4
        b, x1, J1 = p.b_packed(), p.x(), p.J()
        p.set_b_packed(b)
        p.evaluate(with_jacobian=True)
        x2, J2 = p.x(), p.J()
    assert np.array_equal(J1.indptr, J2.indptr) and np.array_equal(J1.indices, J2.indices)
    assert np.array_equal(x1, x2)
    assert np.array_equal(J1.data, J2.data)
    assert np.abs(J1.data).max() > 0 and np.isfinite(J1.data).all()
