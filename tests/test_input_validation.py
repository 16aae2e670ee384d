"""The argument checks of mrcal.optimize() / optimizer_callback() that the
reference performs in its Python wrapper before anything is computed
(mrcal-pywrap.c: optimize_validate_args(), :1063-1305). They are host logic:
restated in mrcal_amd/_api.py, exercised here on the CPU (through the binding
over the reference library, so that the good case is known to go through)."""
import numpy as np
import pytest

from mrcal_amd.synthetic import make_calibration_problem, copy_inputs


@pytest.fixture(scope="module")
def good(ref_api):
    oi, _ = make_calibration_problem(ref_api, Ncameras=2, Nframes=3, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=4, object_height_n=3, seed=1)
    oi["do_apply_outlier_rejection"] = False
    ref_api.optimizer_callback(no_factorization=True, **copy_inputs(oi))     # the good case passes
    return oi


def bad(ref_api, good, exc, fragment, **changes):
    oi = copy_inputs(good)
    for k, v in changes.items():
        if v is KeyError: oi.pop(k)
        else:             oi[k] = v
    with pytest.raises(exc) as e:
        ref_api.optimizer_callback(no_factorization=True, **oi)
    assert fragment in str(e.value), str(e.value)


def test_argument_types_and_shapes(ref_api, good):
    bad(ref_api, good, TypeError,    "Required argument 'intrinsics' missing", intrinsics=KeyError)
    bad(ref_api, good, TypeError,    "'lensmodel' must be a string",           lensmodel=3)
    bad(ref_api, good, RuntimeError, "must have dtype",   intrinsics=good["intrinsics"].astype(np.float32))
    bad(ref_api, good, RuntimeError, "must be c-style contiguous", intrinsics=np.asfortranarray(good["intrinsics"]))
    bad(ref_api, good, RuntimeError, "must have exactly", rt_ref_frame=good["rt_ref_frame"].ravel())
    bad(ref_api, good, RuntimeError, "intrinsics.shape[-1] MUST be 8", intrinsics=np.ascontiguousarray(good["intrinsics"][:,:6]))
    bad(ref_api, good, RuntimeError, "Inconsistent Ncameras", imagersizes=good["imagersizes"][:1].copy())
    bad(ref_api, good, RuntimeError, "Inconsistent Nobservations_board",
        indices_frame_camintrinsics_camextrinsics=good["indices_frame_camintrinsics_camextrinsics"][:-1].copy())
    bad(ref_api, good, RuntimeError, "has a non-null value", no_such_thing=np.ones(3))
    bad(ref_api, good, RuntimeError, "legacy alias", frames_rt_toref=good["rt_ref_frame"].copy())


def test_board_requirements(ref_api, good):
    bad(ref_api, good, RuntimeError, "calibration_object_spacing", calibration_object_spacing=0.0)
    bad(ref_api, good, RuntimeError, "calobject_warp MUST be given", calobject_warp=None)


def test_observation_index_order(ref_api, good):
    idx = good["indices_frame_camintrinsics_camextrinsics"]
    name = "indices_frame_camintrinsics_camextrinsics"
    i = idx.copy(); i[0,0] = 7
    bad(ref_api, good, RuntimeError, "iframe_here MUST be in [0,2]", **{name: i})
    i = idx.copy(); i[1,1] = 5
    bad(ref_api, good, RuntimeError, "icam_intrinsics_here MUST be in [0,1]", **{name: i})
    i = idx.copy(); i[1,2] = 3
    bad(ref_api, good, RuntimeError, "icam_extrinsics_here MUST be in [-1,0]", **{name: i})
    i = idx.copy(); i[[2,3]] = i[[3,2]]; i[2,0], i[3,0] = i[3,0], i[2,0]      # cameras of one frame swapped
    bad(ref_api, good, RuntimeError, "monotonically increasing", **{name: i})
    i = idx.copy(); i[2:,0] += 1                                             # a frame skipped
    rt = np.r_[good["rt_ref_frame"], good["rt_ref_frame"][:1]]
    bad(ref_api, good, RuntimeError, "increasing sequentially", rt_ref_frame=rt, **{name: i})
    rt = np.r_[good["rt_ref_frame"], good["rt_ref_frame"][:1]]               # a frame nobody observes
    bad(ref_api, good, RuntimeError, "must cover ALL frames", rt_ref_frame=rt)


def test_points(ref_api, good):
    pts = np.array(((0.1, 0.2, 5.), (0.3, -0.2, 4.), (0., 0., 6.)))
    idx = np.array(((0,0,-1), (0,1,0), (1,0,-1), (1,1,0), (2,0,-1), (2,1,0)), dtype=np.int32)
    obs = np.column_stack((np.full(6, 1000.), np.full(6, 900.), np.ones(6)))
    ok  = dict(points=pts, observations_point=obs, indices_point_camintrinsics_camextrinsics=idx, Npoints_fixed=1)
    oi = copy_inputs(good); oi.update(ok)
    ref_api.optimizer_callback(no_factorization=True, **oi)                  # good
    bad(ref_api, good, RuntimeError, "Npoints_fixed > Npoints makes no sense", **dict(ok, Npoints_fixed=4))
    bad(ref_api, good, RuntimeError, "shouldn't be given", Npoints_fixed=1)
    i = idx.copy(); i[4:,0] = 5
    bad(ref_api, good, RuntimeError, "i_point_here MUST be in [0,2]", **dict(ok, indices_point_camintrinsics_camextrinsics=i))
    bad(ref_api, good, RuntimeError, "Inconsistent Nobservations_point", **dict(ok, observations_point=obs[:-1].copy()))
    i = idx.copy(); i[2:4,0] = 2; i[4:,0] = 2
    bad(ref_api, good, RuntimeError, "one point at a time", **dict(ok, indices_point_camintrinsics_camextrinsics=i))
    i = idx[:4].copy()
    bad(ref_api, good, RuntimeError, "there are gaps", **dict(ok, indices_point_camintrinsics_camextrinsics=i, observations_point=obs[:4].copy()))


def test_optimize_rejects_callback_flags(ref_api, good):
    with pytest.raises(TypeError) as e:
        ref_api.optimize(no_jacobian=True, **copy_inputs(good))
    assert "invalid keyword argument" in str(e.value)


def test_sigint_is_default_during_the_c_call_and_restored(ref_api):
    """python-wrapping-utilities.h:18-32: SIGINT is SIG_DFL while the C code runs (so that Ctrl-C ends a long
    solve) and Python's handler is back afterwards. Checked on the wrapper with the reference's library: no GPU"""
    import signal
    from mrcal_amd._api import _sigint_default
    from mrcal_amd.synthetic import make_calibration_problem
    mine = lambda *a: None
    old = signal.signal(signal.SIGINT, mine)
    try:
        with _sigint_default():
            assert signal.getsignal(signal.SIGINT) == signal.SIG_DFL
        assert signal.getsignal(signal.SIGINT) is mine
        oi, _ = make_calibration_problem(ref_api, Ncameras=1, Nframes=3, lensmodel="LENSMODEL_OPENCV4",
                                         object_width_n=4, object_height_n=4, seed=1)
        ref_api.optimizer_callback(no_factorization=True, **oi)
        assert signal.getsignal(signal.SIGINT) is mine
        # a failing call restores it too
        import pytest
        with pytest.raises(Exception):
            ref_api.optimize(**dict(oi, lensmodel="LENSMODEL_NOSUCH"))
        assert signal.getsignal(signal.SIGINT) is mine
    finally:
        signal.signal(signal.SIGINT, old)
