"""ctypes binding of the mrcal C ABI for the optimize()/optimizer_callback() path.

The SAME binding drives two shared libraries, because libmrcal_amd.so exports
the reference's own C entry points for this path (include/mrcal_amd.h, "drop-in
tier") with the reference's signatures (mrcal.h:374-853, internal.h:99-114):

  - mrcal_amd/libmrcal_amd.so      the product: HIP kernels behind the C ABI
  - any libmrcal.so-compatible lib e.g. the reference's own sources compiled as
                                   a CPU checker. Only tests/bench do that;
                                   nothing in this package loads anything but
                                   libmrcal_amd.so on its own.

This module is plumbing: no numerics happen here.
"""
import ctypes as C
import os
import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int_p    = C.POINTER(C.c_int)


class Lensmodel(C.Structure):
    """mrcal_lensmodel_t: 16 bytes. types.h:131-145"""
    class _U(C.Union):
        class _Cahvore(C.Structure):
            _fields_ = [("linearity", C.c_double)]
        class _Splined(C.Structure):
            _fields_ = [("order", C.c_uint16), ("Nx", C.c_uint16),
                        ("Ny", C.c_uint16),    ("fov_x_deg", C.c_uint16)]
        _fields_ = [("cahvore", _Cahvore), ("splined", _Splined)]
    _anonymous_ = ("u",)
    _fields_ = [("type", C.c_int), ("u", _U)]
assert C.sizeof(Lensmodel) == 16


class ProblemSelections(C.Structure):
    """mrcal_problem_selections_t: 8 one-bit flags in one byte, passed by
    value. Bit order of types.h:283-301"""
    _fields_ = [("bits", C.c_uint8)]
    NAMES = ("do_optimize_intrinsics_core",
             "do_optimize_intrinsics_distortions",
             "do_optimize_extrinsics",
             "do_optimize_frames",
             "do_optimize_calobject_warp",
             "do_apply_regularization",
             "do_apply_outlier_rejection",
             "do_apply_regularization_unity_cam01")

    @classmethod
    def make(cls, **flags):
        bits = 0
        for i, name in enumerate(cls.NAMES):
            if flags.get(name, False):
                bits |= (1 << i)
        unknown = set(flags) - set(cls.NAMES)
        if unknown:
            raise TypeError(f"unknown problem selections: {unknown}")
        return cls(bits)

    def as_dict(self):
        return {name: bool(self.bits & (1 << i)) for i, name in enumerate(self.NAMES)}
assert C.sizeof(ProblemSelections) == 1


class Stats(C.Structure):
    """mrcal_stats_t. types.h:321-344"""
    _fields_ = [("rms_reproj_error__pixels",     C.c_double),
                ("Noutliers_board",              C.c_int),
                ("Noutliers_triangulated_point", C.c_int)]


class CholmodSparse(C.Structure):
    """The head of SuiteSparse's cholmod_sparse; only p,i,x are written by the
    callback (mrcal.c:4461-4463)"""
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t),
                ("p", C.c_void_p), ("i", C.c_void_p), ("nz", C.c_void_p),
                ("x", C.c_void_p), ("z", C.c_void_p),
                ("stype", C.c_int), ("itype", C.c_int), ("xtype", C.c_int),
                ("dtype", C.c_int), ("sorted", C.c_int), ("packed", C.c_int)]


# observation records. types.h:195-263
observation_board_dtype = np.dtype([("icam_intrinsics", np.int32),
                                    ("icam_extrinsics", np.int32),
                                    ("iframe",          np.int32)])
observation_point_dtype = np.dtype([("icam_intrinsics", np.int32),
                                    ("icam_extrinsics", np.int32),
                                    ("i_point",         np.int32)])
# {int,int,bitfield byte(+7 pad), double[3]}: 40 bytes
observation_point_triangulated_dtype = np.dtype({
    "names":   ["icam_intrinsics", "icam_extrinsics", "flags", "px"],
    "formats": [np.int32, np.int32, np.uint8, (np.float64, 3)],
    "offsets": [0, 4, 8, 16],
    "itemsize": 40})
TRIANGULATED_LAST_IN_SET = 1
TRIANGULATED_OUTLIER     = 2


def _ptr(a, ctype=C.c_void_p):
    if a is None:
        return None
    return a.ctypes.data_as(ctype)


class MrcalLib:
    """One loaded shared library exporting the mrcal C ABI of this path"""

    # the 8 integer arguments every state-layout function ends with:
    #   Ncameras_intrinsics, Ncameras_extrinsics, Nframes, Npoints,
    #   Npoints_fixed, Nobservations_board, problem_selections, lensmodel
    _STATE_ARGS = [C.c_int]*6 + [ProblemSelections, C.POINTER(Lensmodel)]

    def __init__(self, path):
        if not os.path.exists(path):
            raise OSError(f"{path} does not exist")
        self.path = path
        self.lib  = C.CDLL(path)
        L = self.lib

        def sig(name, restype, argtypes):
            f = getattr(L, name)
            f.restype  = restype
            f.argtypes = argtypes
            return f

        sig("mrcal_lensmodel_from_name",  C.c_bool, [C.POINTER(Lensmodel), C.c_char_p])
        sig("mrcal_lensmodel_num_params", C.c_int,  [C.POINTER(Lensmodel)])
        sig("mrcal_num_intrinsics_optimization_params", C.c_int, [ProblemSelections, C.POINTER(Lensmodel)])

        sig("mrcal_num_states", C.c_int, self._STATE_ARGS)
        for what in ("intrinsics", "extrinsics", "frames", "points"):
            sig(f"mrcal_state_index_{what}", C.c_int, [C.c_int] + self._STATE_ARGS)
        sig("mrcal_state_index_calobject_warp", C.c_int, self._STATE_ARGS)
        sig("mrcal_num_states_intrinsics",     C.c_int, [C.c_int, ProblemSelections, C.POINTER(Lensmodel)])
        sig("mrcal_num_states_extrinsics",     C.c_int, [C.c_int, ProblemSelections])
        sig("mrcal_num_states_frames",         C.c_int, [C.c_int, ProblemSelections])
        sig("mrcal_num_states_points",         C.c_int, [C.c_int, C.c_int, ProblemSelections])
        sig("mrcal_num_states_calobject_warp", C.c_int, [ProblemSelections, C.c_int])

        sig("mrcal_pack_solver_state_vector",   None, [c_double_p] + self._STATE_ARGS)
        sig("mrcal_unpack_solver_state_vector", None, [c_double_p] + self._STATE_ARGS)

        sig("mrcal_measurement_index_boards", C.c_int, [C.c_int]*5)
        sig("mrcal_num_measurements_boards",  C.c_int, [C.c_int]*3)
        sig("mrcal_measurement_index_points", C.c_int, [C.c_int]*5)
        sig("mrcal_num_measurements_points",  C.c_int, [C.c_int])
        sig("mrcal_measurement_index_points_triangulated", C.c_int,
            [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int])
        sig("mrcal_num_measurements_points_triangulated", C.c_int, [C.c_void_p, C.c_int])
        sig("mrcal_measurement_index_regularization", C.c_int,
            [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_int]*7 + [ProblemSelections, C.POINTER(Lensmodel)])
        sig("mrcal_num_measurements_regularization", C.c_int, self._STATE_ARGS)
        sig("mrcal_num_measurements", C.c_int,
            [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_int]*5 +
            [ProblemSelections, C.POINTER(Lensmodel)])
        sig("_mrcal_num_j_nonzero", C.c_int,
            [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_int]*5 +
            [C.c_void_p, C.c_void_p, ProblemSelections, C.POINTER(Lensmodel)])
        sig("mrcal_corresponding_icam_extrinsics", C.c_bool,
            [c_int_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p])
        sig("mrcal_decode_observation_indices_points_triangulated", C.c_bool,
            [c_int_p]*6 + [C.c_int, C.c_void_p, C.c_int])

        common = [
            # intrinsics, rt_cam_ref, rt_ref_frame, points, calobject_warp
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            # Ncameras_intrinsics, Ncameras_extrinsics, Nframes, Npoints, Npoints_fixed
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            # observations_board, observations_point, Nobservations_board, Nobservations_point
            C.c_void_p, C.c_void_p, C.c_int, C.c_int,
            # observations_point_triangulated, Nobservations_point_triangulated
            C.c_void_p, C.c_int,
            # observations_board_pool, observations_point_pool
            C.c_void_p, C.c_void_p,
            # lensmodel, imagersizes, problem_selections, problem_constants
            C.POINTER(Lensmodel), C.c_void_p, ProblemSelections, C.c_void_p,
            # calibration_object_spacing, width_n, height_n, verbose
            C.c_double, C.c_int, C.c_int, C.c_bool ]
        sig("mrcal_optimizer_callback", C.c_bool,
            [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(CholmodSparse)] + common)
        sig("mrcal_optimize", Stats,
            [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + common + [C.c_bool])

        if hasattr(L, "_mrcal_drt_cross_reprojection__dbpacked"):
            sig("_mrcal_drt_cross_reprojection__dbpacked", C.c_bool,
                [C.c_void_p, C.c_int, C.c_int]*4 + [C.c_int, C.c_void_p, C.c_int, C.POINTER(CholmodSparse)] +
                [C.c_int]*7 + [C.POINTER(Lensmodel), ProblemSelections, C.c_int, C.c_int])

    def has_symbol(self, name):
        return hasattr(self.lib, name)

    def lensmodel(self, name):
        m = Lensmodel()
        if not self.lib.mrcal_lensmodel_from_name(C.byref(m), name.encode()):
            raise RuntimeError(f"Couldn't parse 'lensmodel' argument '{name}'. "
                               "Is it a string of a known lens model?")
        return m
