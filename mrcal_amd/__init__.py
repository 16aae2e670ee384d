"""mrcal_amd: MI355X-native implementation of mrcal's optimize() /
optimizer_callback() hot path.

    import mrcal_amd as mrcal
    stats          = mrcal.optimize(**optimization_inputs)
    b, x, J, fact  = mrcal.optimizer_callback(**optimization_inputs)
    i              = mrcal.state_index_frames(3, **optimization_inputs)

The names, keyword arguments and return values are those of the reference's
mrcal._mrcal module for this path (mrcal-pywrap.c:4501-4546). All of the compute
happens in mrcal_amd/libmrcal_amd.so (HIP, gfx950); there is NO CPU fallback:
without the library the import fails, without a GPU the compute entry points
raise.
"""
import os as _os

# ONE HIP runtime per process. torch ships its own libamdhip64 (soname
# libamdhip64.so.7, like /opt/rocm's); whichever is loaded first serves
# libmrcal_amd.so too, but only if torch comes first: loaded after us, torch
# brings up a second runtime that finds no GPU. The multi-GPU path needs torch
# (RCCL via torch.distributed), so load it before the library
try:
    import torch as _torch  # noqa: F401
except ImportError:         # the C ABI and the single-GPU path do not need it
    _torch = None

from ._cabi import MrcalLib as _MrcalLib
from ._api  import Api as _Api, optimization_inputs_known_keys as _optimization_inputs_known_keys

# (MRCAL_AMD_LIB: another build of the same library - the measurement build libmrcal_amd_dev.so of csrc/build.sh, dev tools only)
_libpath = _os.environ.get("MRCAL_AMD_LIB") or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmrcal_amd.so")
if not _os.path.exists(_libpath):
    raise ImportError(
        f"{_libpath} is missing. Build it with mrcal_amd/csrc/build.sh (or "
        "python -c 'import __graft_entry__ as g; g.build()' at the repo root). "
        "mrcal_amd has no CPU fallback")

_lib = _MrcalLib(_libpath)
_api = _Api(_lib)

optimize                              = _api.optimize
optimizer_callback                    = _api.optimizer_callback
drt_cross_reprojection__dbpacked      = _api.drt_cross_reprojection__dbpacked
state_index_intrinsics                = _api.state_index_intrinsics
state_index_extrinsics                = _api.state_index_extrinsics
state_index_frames                    = _api.state_index_frames
state_index_points                    = _api.state_index_points
state_index_calobject_warp            = _api.state_index_calobject_warp
num_states                            = _api.num_states
num_states_intrinsics                 = _api.num_states_intrinsics
num_states_extrinsics                 = _api.num_states_extrinsics
num_states_frames                     = _api.num_states_frames
num_states_points                     = _api.num_states_points
num_states_calobject_warp             = _api.num_states_calobject_warp
num_intrinsics_optimization_params    = _api.num_intrinsics_optimization_params
measurement_index_boards              = _api.measurement_index_boards
measurement_index_points              = _api.measurement_index_points
measurement_index_points_triangulated = _api.measurement_index_points_triangulated
measurement_index_regularization      = _api.measurement_index_regularization
num_measurements                      = _api.num_measurements
num_measurements_boards               = _api.num_measurements_boards
num_measurements_points               = _api.num_measurements_points
num_measurements_points_triangulated  = _api.num_measurements_points_triangulated
num_measurements_regularization       = _api.num_measurements_regularization
corresponding_icam_extrinsics         = _api.corresponding_icam_extrinsics
decode_observation_indices_points_triangulated = _api.decode_observation_indices_points_triangulated
pack_state                            = _api.pack_state
unpack_state                          = _api.unpack_state
lensmodel_num_params                  = _api.lensmodel_num_params
project                               = _api.project
unproject                             = _api.unproject

from ._factorization import CHOLMOD_factorization, _Jt_x, _A_Jt_J_At, _A_Jt_J_At__2

# the callers either side of the path: the on-disk format of
# a calibration, host-side pose arithmetic, the seeding. As in the reference, the class
# takes the name of its module
from .poseutils import (identity_R, identity_r, identity_Rt, identity_rt, R_from_r, r_from_R, Rt_from_rt, rt_from_Rt,
                        invert_R, invert_Rt, invert_rt, compose_Rt, compose_rt, compose_r,
                        rotate_point_R, rotate_point_r, transform_point_Rt, transform_point_rt, close_contour)
from .cameramodel import cameramodel, CameramodelParseException
from .calibration import (ref_calibration_object, align_procrustes_points_Rt01, traverse_sensor_links,
                          estimate_monocular_calobject_poses_Rt_tocam, estimate_joint_frame_poses, seed_stereographic)


def gpu_available():
    """True if a HIP device is visible to libmrcal_amd.so"""
    import ctypes
    f = _lib.lib.mrcal_amd_device_count
    f.restype = ctypes.c_int
    return f() > 0


def set_optimize_jacobian_stream(stream):
    """optimize() returns no Jacobian and its problem does not outlive the call, so by default its steps do not
    stream the CSR values of J to HBM (the same results to the bit: include/mrcal_amd.h, round 6). True: they do,
    as the benchmark's metric defines a step. Returns the previous setting"""
    import ctypes
    f = _lib.lib.mrcal_amd_set_optimize_jacobian_stream
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_int]
    return bool(f(1 if stream else 0))


def set_test_hook(name, value):
    """For the tests (include/mrcal_amd.h, mrcal_amd_set_test_hook): force a path of the big camera block's
    factorization that a solve takes by itself only when a later point outgrows what its first point needed.
    Returns the previous value"""
    import ctypes
    f = _lib.lib.mrcal_amd_set_test_hook
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]
    r = f(name.encode(), int(value))
    if r < 0: raise ValueError(f"unknown test hook {name!r}")
    return r
