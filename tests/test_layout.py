"""State/measurement layout, Nnz and pack/unpack: pure integer (and
one-multiply) host logic. Must agree BIT-EXACTLY with the reference
(mrcal.c:337-882, 3288-3880). Checked against the reference's own code
(oracle/_ref) over a sweep of problem shapes and do_optimize_* combinations,
plus the known sizes of the benchmark configurations."""
import ctypes as C
import itertools
import numpy as np
import pytest

from mrcal_amd._cabi import ProblemSelections, observation_board_dtype, observation_point_dtype, _ptr

LENSMODELS = ("LENSMODEL_PINHOLE", "LENSMODEL_STEREOGRAPHIC", "LENSMODEL_LONLAT", "LENSMODEL_LATLON",
              "LENSMODEL_OPENCV4", "LENSMODEL_OPENCV5", "LENSMODEL_OPENCV8", "LENSMODEL_OPENCV12",
              "LENSMODEL_CAHVOR", "LENSMODEL_CAHVORE_linearity=0.37",
              "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=200",
              "LENSMODEL_SPLINED_STEREOGRAPHIC_order=2_Nx=30_Ny=20_fov_x_deg=150")


def test_known_sizes_northstar(amd):
    """8 cameras x 1000 frames x 10x10, OPENCV8, everything optimized:
    Nstate 6140, Nmeas 1,600,080, Nnz 37,200,080 (measured on the reference,
    SURVEY.md section 6)"""
    Ncam, Nf = 8, 1000
    idx = np.zeros((Ncam*Nf,3), dtype=np.int32)
    idx[:,0] = np.repeat(np.arange(Nf), Ncam)
    idx[:,1] = np.tile(np.arange(Ncam), Nf)
    idx[:,2] = idx[:,1] - 1
    kw = dict(lensmodel="LENSMODEL_OPENCV8", Ncameras_intrinsics=Ncam, Ncameras_extrinsics=Ncam-1,
              Nframes=Nf, Nobservations_board=Ncam*Nf,
              calibration_object_width_n=10, calibration_object_height_n=10)
    assert amd.num_states(**kw)       == 6140
    assert amd.num_measurements(**kw) == 1600080
    assert amd.state_index_frames(3, **kw)      == 8*12 + 7*6 + 18
    assert amd.state_index_calobject_warp(**kw) == 6138
    assert amd.measurement_index_regularization(**kw) == 1600000

    api = amd._api
    sel = ProblemSelections.make(**{n: True for n in ProblemSelections.NAMES[:6]})
    lm  = api.lib.lensmodel("LENSMODEL_OPENCV8")
    c_board = np.empty((Ncam*Nf,), dtype=observation_board_dtype)
    c_board["iframe"], c_board["icam_intrinsics"], c_board["icam_extrinsics"] = idx.T
    nnz = api.clib._mrcal_num_j_nonzero(Ncam*Nf, 0, None, 0, 10, 10, Ncam, Ncam-1, Nf, 0, 0,
                                        _ptr(c_board), None, sel, C.byref(lm))
    assert nnz == 37200080


@pytest.mark.parametrize("lensmodel", LENSMODELS)
def test_layout_matches_reference(amd_api, ref_api, lensmodel):
    rng = np.random.RandomState(1)
    lm_a = amd_api.lib.lensmodel(lensmodel)
    lm_r = ref_api.lib.lensmodel(lensmodel)
    assert bytes(lm_a) == bytes(lm_r)
    assert amd_api.clib.mrcal_lensmodel_num_params(C.byref(lm_a)) == \
           ref_api.clib.mrcal_lensmodel_num_params(C.byref(lm_r))

    shapes = [ (1,0,1,0,0,1,0),  (2,1,3,3,1,4,5),  (4,3,50,0,0,200,0),
               (3,3,7,5,0,15,9), (2,5,1,4,4,6,3),  (1,0,0,6,2,0,8),  (5,4,9,0,0,0,0) ]
    for (Nci,Nce,Nf,Np,Npf,Nob,Nop) in shapes:
        W,H = (7,5) if Nob else (0,0)
        c_board = np.zeros((Nob,), dtype=observation_board_dtype)
        c_board["icam_intrinsics"] = rng.randint(0, Nci, size=Nob)
        c_board["icam_extrinsics"] = rng.randint(-1, Nce, size=Nob) if Nce else -1
        c_board["iframe"]          = np.sort(rng.randint(0, max(Nf,1), size=Nob))
        c_point = np.zeros((Nop,), dtype=observation_point_dtype)
        c_point["icam_intrinsics"] = rng.randint(0, Nci, size=Nop)
        c_point["icam_extrinsics"] = rng.randint(-1, Nce, size=Nop) if Nce else -1
        c_point["i_point"]         = rng.randint(0, max(Np,1), size=Nop)

        for bits in itertools.product((False,True), repeat=8):
            if rng.rand() < 0.75 and any(bits):   # subsample the 256 combinations
                continue
            sel = ProblemSelections.make(**dict(zip(ProblemSelections.NAMES, bits)))
            state = (Nci,Nce,Nf,Np,Npf,Nob,sel)
            def both(name, *args):
                a = getattr(amd_api.clib, name)(*[x if x is not Ellipsis else C.byref(lm_a) for x in args])
                r = getattr(ref_api.clib, name)(*[x if x is not Ellipsis else C.byref(lm_r) for x in args])
                assert a == r, f"{name}{args}: ours {a}, reference {r} ({lensmodel}, sel={sel.as_dict()})"
                return a
            Nstate = both("mrcal_num_states", *state, ...)
            both("mrcal_num_intrinsics_optimization_params", sel, ...)
            both("mrcal_num_states_intrinsics", Nci, sel, ...)
            both("mrcal_num_states_extrinsics", Nce, sel)
            both("mrcal_num_states_frames", Nf, sel)
            both("mrcal_num_states_points", Np, Npf, sel)
            both("mrcal_num_states_calobject_warp", sel, Nob)
            both("mrcal_state_index_calobject_warp", *state, ...)
            for i in (-1, 0, 1, 2, 100):
                both("mrcal_state_index_intrinsics", i, *state, ...)
                both("mrcal_state_index_extrinsics", i, *state, ...)
                both("mrcal_state_index_frames",     i, *state, ...)
                both("mrcal_state_index_points",     i, *state, ...)
                both("mrcal_measurement_index_boards", i, Nob, Nop, W, H)
                both("mrcal_measurement_index_points", i, Nob, Nop, W, H)
            both("mrcal_num_measurements_boards", Nob, W, H)
            both("mrcal_num_measurements_points", Nop)
            both("mrcal_num_measurements_regularization", *state, ...)
            both("mrcal_measurement_index_regularization", None, 0, W, H, Nci,Nce,Nf,Np,Npf,Nob,Nop, sel, ...)
            both("mrcal_num_measurements", Nob, Nop, None, 0, W, H, Nci,Nce,Nf,Np,Npf, sel, ...)
            both("_mrcal_num_j_nonzero", Nob, Nop, None, 0, W, H, Nci,Nce,Nf,Np,Npf,
                 _ptr(c_board), _ptr(c_point), sel, ...)

            # pack/unpack: one division / one multiplication per element: bit-exact
            if Nstate > 0:
                b0 = rng.randn(Nstate)
                for f in ("mrcal_pack_solver_state_vector", "mrcal_unpack_solver_state_vector"):
                    ba, br = b0.copy(), b0.copy()
                    getattr(amd_api.clib, f)(ba.ctypes.data_as(C.POINTER(C.c_double)), *state, C.byref(lm_a))
                    getattr(ref_api.clib, f)(br.ctypes.data_as(C.POINTER(C.c_double)), *state, C.byref(lm_r))
                    assert np.array_equal(ba, br), f


def test_lensmodel_name_errors(amd_api, ref_api):
    for name in ("LENSMODEL_OPENCV8x", "LENSMODEL_CAHVORE", "LENSMODEL_CAHVORE_linearity=",
                 "LENSMODEL_SPLINED_STEREOGRAPHIC", "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11",
                 "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=200x", "nonsense", ""):
        from mrcal_amd._cabi import Lensmodel
        ma, mr = Lensmodel(), Lensmodel()
        ra = amd_api.clib.mrcal_lensmodel_from_name(C.byref(ma), name.encode())
        rr = ref_api.clib.mrcal_lensmodel_from_name(C.byref(mr), name.encode())
        assert ra == rr == False
        assert ma.type == mr.type, name


def test_triangulated_decode(amd_api, ref_api):
    from mrcal_amd._cabi import observation_point_triangulated_dtype, TRIANGULATED_LAST_IN_SET
    sets = (2, 5, 3, 2, 4)
    obs = np.zeros((sum(sets),), dtype=observation_point_triangulated_dtype)
    i = 0
    for n in sets:
        i += n
        obs["flags"][i-1] = TRIANGULATED_LAST_IN_SET
    N = len(obs)
    na = amd_api.clib.mrcal_num_measurements_points_triangulated(_ptr(obs), N)
    nr = ref_api.clib.mrcal_num_measurements_points_triangulated(_ptr(obs), N)
    assert na == nr == sum(n*(n-1)//2 for n in sets)
    for ip in range(len(sets)+1):
        assert amd_api.clib.mrcal_measurement_index_points_triangulated(ip, 3, 2, _ptr(obs), N, 4, 5) == \
               ref_api.clib.mrcal_measurement_index_points_triangulated(ip, 3, 2, _ptr(obs), N, 4, 5)
    for m in range(na+2):
        outs = []
        for api in (amd_api, ref_api):
            v = [C.c_int(-7) for _ in range(6)]
            ok = api.clib.mrcal_decode_observation_indices_points_triangulated(
                *[C.byref(x) for x in v], m, _ptr(obs), N)
            outs.append((bool(ok),) + (tuple(x.value for x in v) if ok else ()))
        assert outs[0] == outs[1], (m, outs)


def test_python_decode_observation_indices_points_triangulated(amd_api, ref_api):
    """the Python-level wrapper (mrcal-pywrap.c:3336-3420) over the same C function"""
    idx = np.array(((0,0,-1), (0,1,0), (0,2,1),  (1,0,-1), (1,2,1),  (2,0,-1), (2,1,0), (2,2,1), (2,3,2)), dtype=np.int32)
    kw = dict(indices_point_triangulated_camintrinsics_camextrinsics = idx,
              observations_point_triangulated = np.zeros((len(idx),3)))
    N = amd_api.num_measurements_points_triangulated(**kw)
    assert N == ref_api.num_measurements_points_triangulated(**kw) == 3 + 1 + 6
    for m in range(N):
        d = amd_api.decode_observation_indices_points_triangulated(m, **kw)
        assert d == ref_api.decode_observation_indices_points_triangulated(m, **kw)
        assert set(d) == {"iobservation0", "iobservation1", "iobservation_point0",
                          "Nobservations_this_point", "Nmeasurements_this_point", "ipoint"}
    assert amd_api.decode_observation_indices_points_triangulated(4, **kw)["ipoint"] == 2
    with pytest.raises(RuntimeError):
        amd_api.decode_observation_indices_points_triangulated(N, **kw)
    with pytest.raises(RuntimeError):
        amd_api.decode_observation_indices_points_triangulated(0)


def test_python_helpers_match_reference_test_values(amd):
    """the values test/test-basic-calibration.py:168-232 asserts for its
    problem: 4 cameras, 50 frames, 10x9 board, OPENCV4"""
    Ncam, Nf = 4, 50
    idx = np.zeros((Ncam*Nf,3), dtype=np.int32)
    idx[:,0] = np.repeat(np.arange(Nf), Ncam)
    idx[:,1] = np.tile(np.arange(Ncam), Nf)
    idx[:,2] = idx[:,1] - 1
    oi = dict(intrinsics=np.zeros((Ncam,8)), rt_cam_ref=np.zeros((Ncam-1,6)),
              rt_ref_frame=np.zeros((Nf,6)), observations_board=np.zeros((Ncam*Nf,9,10,3)),
              indices_frame_camintrinsics_camextrinsics=idx,
              lensmodel="LENSMODEL_OPENCV4", imagersizes=np.zeros((Ncam,2),dtype=np.int32),
              calobject_warp=np.zeros((2,)), calibration_object_spacing=0.1,
              do_optimize_intrinsics_core=True, do_optimize_intrinsics_distortions=True,
              do_optimize_extrinsics=True, do_optimize_frames=True,
              do_optimize_calobject_warp=True, do_apply_regularization=True)
    Nintr = 8
    assert amd.state_index_intrinsics(2, **oi) == 8*2
    assert amd.num_states_intrinsics(**oi)     == 8*Ncam
    assert amd.num_intrinsics_optimization_params(**oi) == 8
    assert amd.state_index_extrinsics(2, **oi) == 8*Ncam + 6*2
    assert amd.num_states_extrinsics(**oi)     == 6*(Ncam-1)
    assert amd.state_index_frames(2, **oi)     == 8*Ncam + 6*(Ncam-1) + 6*2
    assert amd.num_states_frames(**oi)         == 6*Nf
    assert amd.state_index_points(2, **oi)     is None
    assert amd.num_states_points(**oi)         == 0
    assert amd.state_index_calobject_warp(**oi) == 8*Ncam + 6*(Ncam-1) + 6*Nf
    assert amd.num_states_calobject_warp(**oi) == 2
    assert amd.num_states(**oi)                == 8*Ncam + 6*(Ncam-1) + 6*Nf + 2
    assert amd.measurement_index_boards(2, **oi) == 10*9*2*2
    assert amd.num_measurements_boards(**oi)   == 10*9*2*Nf*Ncam
    assert amd.measurement_index_points(2, **oi) is None
    assert amd.num_measurements_points(**oi)   == 0
    assert amd.measurement_index_regularization(**oi) == 10*9*2*Nf*Ncam
    assert amd.num_measurements_regularization(**oi)  == Ncam*(4+2)
    assert amd.num_measurements(**oi) == 10*9*2*Nf*Ncam + Ncam*6
    assert amd.corresponding_icam_extrinsics(0, **oi) == -1
    assert amd.corresponding_icam_extrinsics(3, **oi) == 2
    b = np.arange(2*amd.num_states(**oi), dtype=float).reshape(2,-1) + 1.
    b0 = b.copy()
    amd.pack_state(b, **oi); amd.unpack_state(b, **oi)
    np.testing.assert_allclose(b, b0, rtol=1e-15)
