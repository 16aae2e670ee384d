"""Host-side mirror of the reference's Python interface to the optimizer:
optimize(), optimizer_callback(), the state_index_*/num_states_*/
measurement_index_*/num_measurements_* helpers, pack_state()/unpack_state().

Same names, keyword arguments, defaults, in-place-update semantics and error
behaviour as the functions mrcal-pywrap.c registers (mrcal-pywrap.c:4501-4546;
argument tables :890-937; validation :976-1244; defaults :1447-1457;
marshalling :1557-2010; index helpers :2178-3594). What is different is what is
underneath: a C-ABI shared library (see _cabi.py) whose compute entry points
launch HIP kernels.
"""
import ctypes as C
import numpy as np

from . import _cabi
from ._cabi import (ProblemSelections, CholmodSparse, Stats,
                    observation_board_dtype, observation_point_dtype,
                    observation_point_triangulated_dtype, _ptr)

# mrcal-pywrap.c:4554-4654
optimization_inputs_known_keys = frozenset((
    "intrinsics", "lensmodel", "imagersizes",
    "extrinsics_rt_fromref", "frames_rt_toref",
    "rt_cam_ref", "rt_ref_frame", "points",
    "observations_board", "indices_frame_camintrinsics_camextrinsics",
    "observations_point", "indices_point_camintrinsics_camextrinsics",
    "observations_point_triangulated",
    "indices_point_triangulated_camintrinsics_camextrinsics",
    "observed_pixel_uncertainty", "calobject_warp", "Npoints_fixed",
    "do_optimize_intrinsics_core", "do_optimize_intrinsics_distortions",
    "do_optimize_extrinsics", "do_optimize_frames", "do_optimize_calobject_warp",
    "calibration_object_spacing", "verbose",
    "do_apply_regularization", "do_apply_regularization_unity_cam01",
    "do_apply_outlier_rejection", "imagepaths",
    # optimizer_callback() extras
    "no_jacobian", "no_factorization"))


def _is_null(x):
    return x is None


def _check_layout(name, a, dtype, dims):
    """No silent casting: exact dtype, C-contiguous, matching shape
    (python-wrapping-utilities.h:67-114)"""
    if a is None:
        return
    if not isinstance(a, np.ndarray):
        raise RuntimeError(f"'{name}' must be a numpy array (or None)")
    if len(dims) != a.ndim:
        raise RuntimeError(f"'{name}' must have exactly {len(dims)} dims; got {a.ndim}")
    for i, d in enumerate(dims):
        if d >= 0 and d != a.shape[i]:
            raise RuntimeError(f"'{name}' must have dimensions '{dims}' where <0 means 'any'. "
                               f"Dims {i} got {a.shape[i]} instead")
    if a.size > 0:
        if a.dtype != dtype:
            raise RuntimeError(f"'{name}' must have dtype: {np.dtype(dtype).name}; got {a.dtype.name}")
        if not a.flags["C_CONTIGUOUS"]:
            raise RuntimeError(f"'{name}' must be c-style contiguous")


class _Problem:
    """The validated, marshalled arguments of one optimize()/optimizer_callback()
    call: numpy views the C ABI can take pointers of"""
    pass


class _sigint_default:
    """Ctrl-C during the C call: the reference's wrapper puts SIGINT back to SIG_DFL for the duration of
    optimize() / optimizer_callback() / drt_cross_reprojection__dbpacked() and restores Python's handler afterwards
    (python-wrapping-utilities.h:18-32, mrcal-pywrap.c:1581, 2139): Python's own handler only sets a flag that
    nothing reads while C code runs, so a long solve could not be interrupted. Same here (ctypes holds the
    interpreter the same way). Signal handlers belong to the main thread: elsewhere this does nothing"""
    def __enter__(self):
        import signal, threading
        self.old = None
        if threading.current_thread() is threading.main_thread():
            try:
                self.old = signal.signal(signal.SIGINT, signal.SIG_DFL)
            except (ValueError, OSError):
                self.old = None
        return self
    def __exit__(self, *a):
        import signal
        if self.old is not None:
            signal.signal(signal.SIGINT, self.old)
        return False


class Api:
    def __init__(self, lib):
        self.lib  = lib       # a _cabi.MrcalLib
        self.clib = lib.lib

    # ------------------------------------------------------------------ #
    # argument ingestion                                                  #
    # ------------------------------------------------------------------ #
    @staticmethod
    def _delete_unknown(kwargs):
        """Forward compatibility: unknown keys are dropped if they are falsy /
        empty, an error otherwise (mrcal-pywrap.c:1491-1555)"""
        out = {}
        for k, v in kwargs.items():
            if k in optimization_inputs_known_keys:
                out[k] = v
                continue
            if isinstance(v, np.ndarray):
                nonnull = v.size != 0
            else:
                nonnull = bool(v)
            if nonnull:
                raise RuntimeError(f"optimization_inputs key '{k}' has a non-null value. "
                                   "Unsupported in this version of mrcal")
        return out

    def _ingest(self, kwargs, callback):
        kw = self._delete_unknown(kwargs)
        if not callback:
            for k in ("no_jacobian", "no_factorization"):
                if k in kw:
                    raise TypeError(f"'{k}' is an invalid keyword argument for mrcal.optimize()")
        for k in ("intrinsics", "lensmodel", "imagersizes"):
            if k not in kw:
                raise TypeError(f"Required argument '{k}' missing")

        def get(name):
            v = kw.get(name, None)
            # the "ERROR:..." poison strings of cameramodel.py mean None
            if isinstance(v, str) and v.startswith("ERROR:"):
                return None
            return v

        p = _Problem()
        lensmodel_name = kw["lensmodel"]
        if not isinstance(lensmodel_name, str):
            raise TypeError("'lensmodel' must be a string")

        intrinsics  = get("intrinsics")
        imagersizes = get("imagersizes")
        rt_cam_ref   = get("rt_cam_ref")
        rt_ref_frame = get("rt_ref_frame")
        old_rt_cam_ref   = get("extrinsics_rt_fromref")
        old_rt_ref_frame = get("frames_rt_toref")

        def renamed(old, name_old, new, name_new):
            if new is not None and new.size and old is not None and old.size:
                raise RuntimeError(f"Both {name_old} and {name_new} are given: "
                                   "the former is a legacy alias for the latter")
            return old if new is None else new
        rt_cam_ref   = renamed(old_rt_cam_ref,   "extrinsics_rt_fromref", rt_cam_ref,   "rt_cam_ref")
        rt_ref_frame = renamed(old_rt_ref_frame, "frames_rt_toref",       rt_ref_frame, "rt_ref_frame")

        def or_empty(a, shape, dtype):
            return np.zeros(shape, dtype=dtype) if a is None else a

        points             = get("points")
        observations_board = get("observations_board")
        idx_board          = get("indices_frame_camintrinsics_camextrinsics")
        observations_point = get("observations_point")
        idx_point          = get("indices_point_camintrinsics_camextrinsics")
        observations_tri   = get("observations_point_triangulated")
        idx_tri            = get("indices_point_triangulated_camintrinsics_camextrinsics")
        calobject_warp     = get("calobject_warp")

        _check_layout("intrinsics",   intrinsics,   np.float64, (-1,-1))
        _check_layout("imagersizes",  imagersizes,  np.int32,   (-1,2))
        _check_layout("rt_cam_ref",   rt_cam_ref,   np.float64, (-1,6))
        _check_layout("rt_ref_frame", rt_ref_frame, np.float64, (-1,6))
        _check_layout("points",       points,       np.float64, (-1,3))
        _check_layout("observations_board", observations_board, np.float64, (-1,-1,-1,3))
        _check_layout("indices_frame_camintrinsics_camextrinsics", idx_board, np.int32, (-1,3))
        _check_layout("observations_point", observations_point, np.float64, (-1,3))
        _check_layout("indices_point_camintrinsics_camextrinsics", idx_point, np.int32, (-1,3))
        _check_layout("observations_point_triangulated", observations_tri, np.float64, (-1,3))
        _check_layout("indices_point_triangulated_camintrinsics_camextrinsics", idx_tri, np.int32, (-1,3))
        _check_layout("calobject_warp", calobject_warp, np.float64, (2,))
        if intrinsics is None:
            raise RuntimeError("'intrinsics' must be an array")

        rt_cam_ref         = or_empty(rt_cam_ref,         (0,6), np.float64)
        rt_ref_frame       = or_empty(rt_ref_frame,       (0,6), np.float64)
        points             = or_empty(points,             (0,3), np.float64)
        observations_board = or_empty(observations_board, (0,179,171,3), np.float64)
        idx_board          = or_empty(idx_board,          (0,3), np.int32)
        observations_point = or_empty(observations_point, (0,3), np.float64)
        idx_point          = or_empty(idx_point,          (0,3), np.int32)
        observations_tri   = or_empty(observations_tri,   (0,3), np.float64)
        idx_tri            = or_empty(idx_tri,            (0,3), np.int32)
        imagersizes        = or_empty(imagersizes,        (0,2), np.int32)

        Ncameras_intrinsics = intrinsics.shape[0]
        Ncameras_extrinsics = rt_cam_ref.shape[0]
        Nframes             = rt_ref_frame.shape[0]
        Npoints             = points.shape[0]
        Nobservations_board = observations_board.shape[0]
        Nobservations_point = observations_point.shape[0]
        Nobservations_tri   = observations_tri.shape[0]
        Npoints_fixed       = int(kw.get("Npoints_fixed", 0))
        spacing             = float(kw.get("calibration_object_spacing", -1.0))

        if imagersizes.shape[0] != Ncameras_intrinsics:
            raise RuntimeError(f"Inconsistent Ncameras: 'intrinsics' says {Ncameras_intrinsics}, "
                               f"'imagersizes' says {imagersizes.shape[0]}")
        if idx_board.shape[0] != Nobservations_board:
            raise RuntimeError(f"Inconsistent Nobservations_board: 'observations_board' says {Nobservations_board}, "
                               f"'indices_frame_camintrinsics_camextrinsics' says {idx_board.shape[0]}")

        def flag(name, default):
            v = kw.get(name, None)
            if v is None:
                return default
            v = int(bool(v)) if not isinstance(v, (int, np.integer)) or v >= 0 else -1
            return v
        do_core  = flag("do_optimize_intrinsics_core",        -1)
        do_dist  = flag("do_optimize_intrinsics_distortions", -1)
        do_ext   = flag("do_optimize_extrinsics",             -1)
        do_frame = flag("do_optimize_frames",                 -1)
        do_warp  = flag("do_optimize_calobject_warp",         -1)

        if Nobservations_board > 0:
            if spacing <= 0.0:
                raise RuntimeError("We have board observations, so calibration_object_spacing "
                                   "MUST be a valid float > 0")
            # note: the unresolved (-1 = auto) flag counts as "true" here, as
            # it does in the reference
            if do_warp and calobject_warp is None:
                raise RuntimeError("do_optimize_calobject_warp is True, so calobject_warp MUST be given "
                                   "as an array to seed the optimization and to receive the results")
        if idx_point.shape[0] != Nobservations_point:
            raise RuntimeError(f"Inconsistent Nobservations_point: 'observations_point...' says {Nobservations_point}, "
                               f"'indices_point_camintrinsics_camextrinsics' says {idx_point.shape[0]}")
        if idx_tri.shape[0] != Nobservations_tri:
            raise RuntimeError("Inconsistent Nobservations_point_triangulated")

        lensmodel = self.lib.lensmodel(lensmodel_name)
        Nlens = self.clib.mrcal_lensmodel_num_params(C.byref(lensmodel))
        if intrinsics.shape[-1] != Nlens:
            raise RuntimeError(f"intrinsics.shape[-1] MUST be {Nlens}. Instead got {intrinsics.shape[-1]}")

        self._validate_indices(idx_board, idx_point, idx_tri,
                               Nframes, Ncameras_intrinsics, Ncameras_extrinsics,
                               Npoints, Npoints_fixed)

        # CONSTRUCT_PROBLEM_SELECTIONS: <0 means "optimize it if there is any"
        sel = ProblemSelections.make(
            do_optimize_intrinsics_core        = bool(do_core  if do_core  >= 0 else Ncameras_intrinsics > 0),
            do_optimize_intrinsics_distortions = bool(do_dist  if do_dist  >= 0 else Ncameras_intrinsics > 0),
            do_optimize_extrinsics             = bool(do_ext   if do_ext   >= 0 else Ncameras_extrinsics > 0),
            do_optimize_frames                 = bool(do_frame if do_frame >= 0 else Nframes > 0),
            do_optimize_calobject_warp         = bool(do_warp  if do_warp  >= 0 else Nobservations_board > 0),
            do_apply_regularization            = bool(kw.get("do_apply_regularization", 1)),
            do_apply_outlier_rejection         = bool(kw.get("do_apply_outlier_rejection", 1)),
            do_apply_regularization_unity_cam01= bool(kw.get("do_apply_regularization_unity_cam01", 0)))

        # (iframe, icam_i, icam_e) rows -> {icam_i, icam_e, iframe} records
        c_board = np.empty((Nobservations_board,), dtype=observation_board_dtype)
        c_board["iframe"]          = idx_board[:,0]
        c_board["icam_intrinsics"] = idx_board[:,1]
        c_board["icam_extrinsics"] = idx_board[:,2]
        c_point = np.empty((Nobservations_point,), dtype=observation_point_dtype)
        c_point["i_point"]         = idx_point[:,0]
        c_point["icam_intrinsics"] = idx_point[:,1]
        c_point["icam_extrinsics"] = idx_point[:,2]

        c_tri = self._fill_triangulated(observations_tri, idx_tri, lensmodel, intrinsics)

        p.lensmodel = lensmodel
        p.lensmodel_name = lensmodel_name
        p.sel = sel
        p.intrinsics = intrinsics
        p.rt_cam_ref = rt_cam_ref
        p.rt_ref_frame = rt_ref_frame
        p.points = points
        p.calobject_warp = calobject_warp
        p.observations_board = observations_board
        p.observations_point = observations_point
        p.c_board = c_board
        p.c_point = c_point
        p.c_tri = c_tri
        p.imagersizes = imagersizes
        p.Ncameras_intrinsics = Ncameras_intrinsics
        p.Ncameras_extrinsics = Ncameras_extrinsics
        p.Nframes = Nframes
        p.Npoints = Npoints
        p.Npoints_fixed = Npoints_fixed
        p.Nobservations_board = Nobservations_board
        p.Nobservations_point = Nobservations_point
        p.Nobservations_tri = Nobservations_tri
        p.spacing = spacing
        p.height_n = observations_board.shape[1] if Nobservations_board > 0 else -1
        p.width_n  = observations_board.shape[2] if Nobservations_board > 0 else -1
        p.verbose = bool(kw.get("verbose", 0))
        p.no_factorization = bool(kw.get("no_factorization", 0))
        p.no_jacobian      = bool(kw.get("no_jacobian", 0)) and p.no_factorization
        return p

    @staticmethod
    def _validate_indices(idx_board, idx_point, idx_tri,
                          Nframes, Ncameras_intrinsics, Ncameras_extrinsics,
                          Npoints, Npoints_fixed):
        """mrcal-pywrap.c:1063-1244, vectorised"""
        name = "indices_frame_camintrinsics_camextrinsics"
        if idx_board.shape[0]:
            f, ci, ce = idx_board[:,0], idx_board[:,1], idx_board[:,2]
            bad = np.nonzero((f < 0) | (f >= Nframes))[0]
            if bad.size:
                raise RuntimeError(f"iframe_here MUST be in [0,{Nframes-1}], instead got {f[bad[0]]} in row {bad[0]} of {name}")
            bad = np.nonzero((ci < 0) | (ci >= Ncameras_intrinsics))[0]
            if bad.size:
                raise RuntimeError(f"icam_intrinsics_here MUST be in [0,{Ncameras_intrinsics-1}], instead got {ci[bad[0]]} in row {bad[0]} of {name}")
            bad = np.nonzero((ce < -1) | (ce >= Ncameras_extrinsics))[0]
            if bad.size:
                raise RuntimeError(f"icam_extrinsics_here MUST be in [-1,{Ncameras_extrinsics-1}], instead got {ce[bad[0]]} in row {bad[0]} of {name}")
            fprev = np.concatenate(((-1,), f[:-1]))
            df    = f - fprev
            bad = np.nonzero(df < 0)[0]
            if bad.size:
                raise RuntimeError(f"iframe_here MUST be monotonically increasing in {name}. Instead row {bad[0]} has iframe_here={f[bad[0]]} after previously seeing iframe_here={fprev[bad[0]]}")
            bad = np.nonzero(df > 1)[0]
            if bad.size:
                raise RuntimeError(f"iframe_here MUST be increasing sequentially in {name}. Instead row {bad[0]} has iframe_here={f[bad[0]]} after previously seeing iframe_here={fprev[bad[0]]}")
            same = np.nonzero(df[1:] == 0)[0] + 1
            bad = same[ci[same] < ci[same-1]]
            if bad.size:
                raise RuntimeError(f"icam_intrinsics_here MUST be monotonically increasing in {name}. Instead row {bad[0]} (frame {f[bad[0]]}) has icam_intrinsics_here={ci[bad[0]]} after previously seeing icam_intrinsics_here={ci[bad[0]-1]}")
            bad = same[ce[same] < ce[same-1]]
            if bad.size:
                raise RuntimeError(f"icam_extrinsics_here MUST be monotonically increasing in {name}. Instead row {bad[0]} (frame {f[bad[0]]}) has icam_extrinsics_here={ce[bad[0]]} after previously seeing icam_extrinsics_here={ce[bad[0]-1]}")
            if f[-1] != Nframes-1:
                raise RuntimeError(f"iframe in {name} must cover ALL frames. Instead the last row of {name} has iframe={f[-1]}, but Nframes={Nframes}")

        if Npoints > 0:
            if Npoints_fixed > Npoints:
                raise RuntimeError(f"I have Npoints=len(points)={Npoints}, but Npoints_fixed={Npoints_fixed}. Npoints_fixed > Npoints makes no sense")
        elif Npoints_fixed:
            raise RuntimeError("No 'points' were given, so it's 'Npoints_fixed' doesn't do anything, and shouldn't be given")

        name = "indices_point_camintrinsics_camextrinsics"
        biggest = -1
        if idx_point.shape[0]:
            ip, ci, ce = idx_point[:,0], idx_point[:,1], idx_point[:,2]
            bad = np.nonzero((ip < 0) | (ip >= Npoints))[0]
            if bad.size:
                raise RuntimeError(f"i_point_here MUST be in [0,{Npoints-1}], instead got {ip[bad[0]]} in row {bad[0]} of {name}")
            bad = np.nonzero((ci < 0) | (ci >= Ncameras_intrinsics))[0]
            if bad.size:
                raise RuntimeError(f"icam_intrinsics_here MUST be in [0,{Ncameras_intrinsics-1}], instead got {ci[bad[0]]} in row {bad[0]} of {name}")
            bad = np.nonzero((ce < -1) | (ce >= Ncameras_extrinsics))[0]
            if bad.size:
                raise RuntimeError(f"icam_extrinsics_here MUST be in [-1,{Ncameras_extrinsics-1}], instead got {ce[bad[0]]} in row {bad[0]} of {name}")
            running = np.maximum.accumulate(np.concatenate(((-1,), ip)))
            bad = np.nonzero(ip > running[:-1] + 1)[0]
            if bad.size:
                raise RuntimeError(f"{name} should contain i_point_here that extend the existing set by one point at a time at most. However row {bad[0]} has i_point_here={ip[bad[0]]} while the biggest-seen-so-far i_point_here={running[bad[0]]}")
            biggest = int(running[-1])
        if biggest != Npoints-1:
            raise RuntimeError(f"{name} should cover all point indices in [0,{Npoints-1}], but there are gaps. The biggest i_point={biggest}")

    def _fill_triangulated(self, observations_tri, idx_tri, lensmodel, intrinsics):
        """(px,py,weight) + (ipoint,icam_i,icam_e) rows -> the C records: the
        pixel unprojected to an observation vector in the camera's coordinates,
        the set boundaries, the outlier bit (mrcal-pywrap.c:1309-1440)"""
        from ._cabi import TRIANGULATED_LAST_IN_SET, TRIANGULATED_OUTLIER
        N = idx_tri.shape[0]
        c_tri = np.zeros((N,), dtype=observation_point_triangulated_dtype)
        if N == 0:
            return c_tri
        ipoint = idx_tri[:,0]
        if ipoint[0] != 0 or np.any(ipoint < 0):
            raise RuntimeError("Error in indices_point_triangulated_camintrinsics_camextrinsics: ipoint must start at 0 and be >= 0")
        d = np.diff(ipoint)
        if np.any((d != 0) & (d != 1)):
            i = int(np.nonzero((d != 0) & (d != 1))[0][0]) + 1
            raise RuntimeError(f"Error in indices_point_triangulated_camintrinsics_camextrinsics[{i}]. All ipoint must be consecutive and monotonic")
        last = np.ones((N,), dtype=bool)
        last[:-1] = d == 1
        counts = np.bincount(ipoint)
        if np.any(counts < 2):
            raise RuntimeError(f"Error in indices_point_triangulated_camintrinsics_camextrinsics. Each point must be observed at least 2 times; point {int(np.argmin(counts))} is not")
        c_tri["icam_intrinsics"] = idx_tri[:,1]
        c_tri["icam_extrinsics"] = idx_tri[:,2]
        flags = np.where(last, TRIANGULATED_LAST_IN_SET, 0).astype(np.uint8)
        flags |= np.where(observations_tri[:,2] <= 0.0, TRIANGULATED_OUTLIER, 0).astype(np.uint8)
        c_tri["flags"] = flags
        if lensmodel is None:
            return c_tri
        # one mrcal_unproject() call per camera
        f = self.clib.mrcal_unproject
        f.restype  = C.c_bool
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        px = np.zeros((N,3))
        for icam in np.unique(idx_tri[:,1]):
            sel_ = np.nonzero(idx_tri[:,1] == icam)[0]
            q = np.ascontiguousarray(observations_tri[sel_,:2])
            v = np.zeros((len(sel_),3))
            intr = np.ascontiguousarray(intrinsics[icam])
            if not f(_ptr(v), _ptr(q), len(sel_), C.byref(lensmodel), _ptr(intr)):
                raise RuntimeError("mrcal_unproject() failed" + self._last_error())
            px[sel_] = v
        c_tri["px"] = px
        return c_tri

    def _common_args(self, p):
        return [
            _ptr(p.intrinsics), _ptr(p.rt_cam_ref), _ptr(p.rt_ref_frame), _ptr(p.points),
            _ptr(p.calobject_warp),
            p.Ncameras_intrinsics, p.Ncameras_extrinsics, p.Nframes, p.Npoints, p.Npoints_fixed,
            _ptr(p.c_board), _ptr(p.c_point), p.Nobservations_board, p.Nobservations_point,
            _ptr(p.c_tri) if p.Nobservations_tri else None, p.Nobservations_tri,
            _ptr(p.observations_board), _ptr(p.observations_point),
            C.byref(p.lensmodel), _ptr(p.imagersizes), p.sel, None,
            p.spacing, p.width_n, p.height_n, p.verbose ]

    def _sizes(self, p):
        tri = _ptr(p.c_tri) if p.Nobservations_tri else None
        Nmeas = self.clib.mrcal_num_measurements(
            p.Nobservations_board, p.Nobservations_point, tri, p.Nobservations_tri,
            p.width_n, p.height_n,
            p.Ncameras_intrinsics, p.Ncameras_extrinsics, p.Nframes,
            p.Npoints, p.Npoints_fixed, p.sel, C.byref(p.lensmodel))
        Nstate = self.clib.mrcal_num_states(
            p.Ncameras_intrinsics, p.Ncameras_extrinsics, p.Nframes,
            p.Npoints, p.Npoints_fixed, p.Nobservations_board, p.sel, C.byref(p.lensmodel))
        return Nstate, Nmeas

    # ------------------------------------------------------------------ #
    # the two compute entry points                                        #
    # ------------------------------------------------------------------ #
    def optimize(self, **kwargs):
        """mrcal.optimize(): solve; updates intrinsics, rt_cam_ref, rt_ref_frame,
        points, calobject_warp and the outlier marks in observations_board IN
        PLACE; returns the stats dict (mrcal-pywrap.c:1809-1888)"""
        # (the reference's Python wrapper always passes check_gradient=false,
        #  mrcal-pywrap.c:1842; the C entry point has it, and so the tests reach it)
        kwargs = dict(kwargs)
        check_gradient = bool(kwargs.pop("_check_gradient", False))
        p = self._ingest(kwargs, callback=False)
        Nstate, Nmeas = self._sizes(p)
        b_packed = np.empty((Nstate,), dtype=np.float64)
        x        = np.empty((Nmeas,),  dtype=np.float64)
        with _sigint_default():
            stats = self.clib.mrcal_optimize(None if check_gradient else _ptr(b_packed), Nstate*8,
                                             None if check_gradient else _ptr(x), Nmeas*8,
                                             *self._common_args(p), check_gradient)
        if stats.rms_reproj_error__pixels < 0.0:
            raise RuntimeError("mrcal.optimize() failed!" + self._last_error())
        # (the C records of the triangulated observations are this wrapper's, as in mrcal-pywrap.c: the outlier bits the
        #  solve left in them do not reach the Python caller. Kept for the tests, which compare them with the reference's)
        self._last_triangulated_flags = p.c_tri["flags"].copy() if p.Nobservations_tri else None
        return dict(rms_reproj_error__pixels     = stats.rms_reproj_error__pixels,
                    Noutliers_board              = stats.Noutliers_board,
                    Noutliers_triangulated_point = stats.Noutliers_triangulated_point,
                    b_packed                     = b_packed,
                    x                            = x)

    def optimizer_callback(self, **kwargs):
        """mrcal.optimizer_callback(): (b_packed, x, J, factorization) at the
        given operating point (mrcal-pywrap.c:1890-2010)"""
        import scipy.sparse
        p = self._ingest(kwargs, callback=True)
        Nstate, Nmeas = self._sizes(p)
        b_packed = np.empty((Nstate,), dtype=np.float64)
        x        = np.empty((Nmeas,),  dtype=np.float64)
        Jt = None
        if not p.no_jacobian:
            tri = _ptr(p.c_tri) if p.Nobservations_tri else None
            Nnz = self.clib._mrcal_num_j_nonzero(
                p.Nobservations_board, p.Nobservations_point, tri, p.Nobservations_tri,
                p.width_n, p.height_n,
                p.Ncameras_intrinsics, p.Ncameras_extrinsics, p.Nframes,
                p.Npoints, p.Npoints_fixed,
                _ptr(p.c_board), _ptr(p.c_point), p.sel, C.byref(p.lensmodel))
            P = np.empty((Nmeas+1,), dtype=np.int32)
            I = np.empty((Nnz,),     dtype=np.int32)
            X = np.empty((Nnz,),     dtype=np.float64)
            Jt = CholmodSparse(nrow=Nstate, ncol=Nmeas, nzmax=Nnz,
                               p=P.ctypes.data, i=I.ctypes.data, x=X.ctypes.data,
                               stype=0, itype=0, xtype=1, dtype=0, sorted=1, packed=1)
        with _sigint_default():
            ok = self.clib.mrcal_optimizer_callback(_ptr(b_packed), Nstate*8, _ptr(x), Nmeas*8,
                                                    C.byref(Jt) if Jt is not None else None,
                                                    *self._common_args(p))
        if not ok:
            raise RuntimeError("mrcal_optimizer_callback() failed!" + self._last_error())
        J = None
        factorization = None
        if Jt is not None:
            J = scipy.sparse.csr_matrix((X, I, P), shape=(Nmeas, Nstate))
            if not p.no_factorization:
                factorization = self._factorization(J, p)
        return b_packed, x, J, factorization

    def drt_cross_reprojection__dbpacked(self, icam_intrinsics=-1, **kwargs):
        """mrcal.drt_cross_reprojection__dbpacked() (mrcal-pywrap.c:2012-2110, 2156-2161): K (6,Nstate) =
        drt_ref_refperturbed/db_packed (icam_intrinsics < 0 or None: the "rrp" form) or
        drt_cam_camperturbed/db_packed of that camera ("ccp"), zero outside the extrinsics / frames / points /
        calobject_warp columns: the cross-reprojection uncertainty's link between a perturbation of the solve
        and the transform that compensates for it (mrcal/model_analysis.py:1379, 1441). One evaluation of the
        Jacobian at the given state, then _mrcal_drt_cross_reprojection__dbpacked() (uncertainty.c:798)"""
        icam = -1 if icam_intrinsics is None else int(icam_intrinsics)
        kwargs = dict(kwargs, no_jacobian=False, no_factorization=True)
        tamper = kwargs.pop("_tamper_with_J", None)      # tests: a malformed Jt must be refused, not walked
        p = self._ingest(kwargs, callback=True)
        if icam >= p.Ncameras_intrinsics:
            raise RuntimeError(f"icam_intrinsics MUST be <0 (if unused) or in [0,Ncameras_intrinsics-1]. "
                               f"got {icam} NOT in [0,{p.Ncameras_intrinsics-1}]")
        b_packed, x, J, _ = self.optimizer_callback(**kwargs)
        Nstate, Nmeas = J.shape[1], J.shape[0]
        if tamper is not None: tamper(J)
        Jt = CholmodSparse(nrow=Nstate, ncol=Nmeas, nzmax=J.nnz,
                           p=J.indptr.ctypes.data, i=J.indices.ctypes.data, x=J.data.ctypes.data,
                           stype=0, itype=0, xtype=1, dtype=0, sorted=1, packed=1)
        s = (p.Ncameras_intrinsics, p.Ncameras_extrinsics, p.Nframes, p.Npoints, p.Npoints_fixed,
             p.Nobservations_board, p.sel, C.byref(p.lensmodel))
        K = np.zeros((6, Nstate), dtype=np.float64)
        def block(i0):
            return (K.ctypes.data + 8*i0 if i0 >= 0 else None), K.strides[0], K.strides[1]
        i_e  = self.clib.mrcal_state_index_extrinsics(0, *s)
        i_f  = self.clib.mrcal_state_index_frames(0, *s)
        i_p  = self.clib.mrcal_state_index_points(0, *s)
        i_cw = self.clib.mrcal_state_index_calobject_warp(*s)
        with _sigint_default():
            ok = self.clib._mrcal_drt_cross_reprojection__dbpacked(
                *block(i_e), *block(i_f), *block(i_p), *block(i_cw),
                icam, _ptr(b_packed), Nstate*8, C.byref(Jt),
                p.Ncameras_intrinsics, p.Ncameras_extrinsics, p.Nframes, p.Npoints, p.Npoints_fixed,
                p.Nobservations_board, p.Nobservations_point,
                C.byref(p.lensmodel), p.sel, p.width_n, p.height_n)
        if not ok:
            raise RuntimeError("_mrcal_drt_cross_reprojection__dbpacked() failed" + self._last_error())
        return K

    def _factorization(self, J, p):
        """the factorization of JtJ that optimizer_callback() returns; with the
        product library the structured GPU solver, told how the state splits"""
        if not self.lib.has_symbol("mrcal_amd_factorization_create"):
            return None
        from ._factorization import CHOLMOD_factorization
        if self.lib.has_symbol("mrcal_amd_factorization_create_from_problem"):
            # from a resident copy of the problem at the same state: its own atomics-free normal equations, and the
            # 466 MB of CSR (at the metric's size) that just came down do not go back up
            from .resident import Problem
            from . import _lib
            if _lib is self.lib:
                # None means "JtJ is singular" and nothing else (mrcal-pywrap.c:1981-1988). Anything else that stops
                # this route - no device memory for a second resident problem, a shard, a Gram past 2^32 doubles -
                # falls through to the factorization of the CSR that was just made, which needs far less
                try:
                    with Problem(_ingested=p) as prob:
                        F = prob.factorization()
                        # (an explicit status, not a search of the error text: ADVICE r5)
                        if F is not None or self.lib.lib.mrcal_amd_factorization_last_status() == 1:
                            return F
                except RuntimeError:
                    pass
        s = (p.Ncameras_intrinsics, p.Ncameras_extrinsics, p.Nframes, p.Npoints, p.Npoints_fixed,
             p.Nobservations_board, p.sel, C.byref(p.lensmodel))
        Ni  = self.clib.mrcal_num_states_intrinsics(p.Ncameras_intrinsics, p.sel, C.byref(p.lensmodel))
        Ne  = self.clib.mrcal_num_states_extrinsics(p.Ncameras_extrinsics, p.sel)
        Nf  = self.clib.mrcal_num_states_frames(p.Nframes, p.sel)
        Np  = self.clib.mrcal_num_states_points(p.Npoints, p.Npoints_fixed, p.sel)
        Nw  = self.clib.mrcal_num_states_calobject_warp(p.sel, p.Nobservations_board)
        try:
            return CHOLMOD_factorization(J, _partition=(Ni + Ne, Nf//6, Np//3, Nw))
        except RuntimeError:
            # a failed factorization is None, not an exception
            # (mrcal-pywrap.c:1981-1988)
            return None

    @staticmethod
    def _broadcast_models(pts, npt, intr, Ni, what):
        """mrcal.project()/unproject() broadcast over the points AND over the
        intrinsics (numpysane prototypes (3,),(Nintrinsics,) / (2,),(Nintrinsics,)):
        -> points (B,npt), intrinsics (B,Ni) flattened to the common leading
        shape, and that shape"""
        pts  = np.asarray(pts,  dtype=np.float64)
        intr = np.asarray(intr, dtype=np.float64)
        if pts.shape[-1] != npt:
            raise RuntimeError(f"{what} must have shape (...,{npt})")
        if intr.ndim < 1 or intr.shape[-1] != Ni:
            raise RuntimeError(f"intrinsics_data must have shape (...,{Ni})")
        lead = np.broadcast_shapes(pts.shape[:-1], intr.shape[:-1])
        pts_b  = np.ascontiguousarray(np.broadcast_to(pts,  lead + (npt,))).reshape(-1, npt)
        if intr.ndim == 1:
            return pts_b, intr.reshape(1, Ni), None, lead
        intr_b = np.ascontiguousarray(np.broadcast_to(intr, lead + (Ni,))).reshape(-1, Ni)
        # the distinct models, and which points go through each
        uniq, inverse = np.unique(intr_b, axis=0, return_inverse=True)
        return pts_b, uniq, inverse.reshape(-1), lead

    def project(self, v, lensmodel, intrinsics_data, get_gradients=False, out=None):
        """q = project(v): camera-frame points (...,3) through a lens model, on
        the GPU (mrcal.project(), mrcal/projections.py:14-110 over
        mrcal_project(), mrcal.h:374-391). Broadcasts over v and over
        intrinsics_data (...,Nintrinsics) like the reference. With get_gradients:
        (q, dq_dv (...,2,3), dq_dintrinsics (...,2,Nintrinsics))"""
        m = self.lib.lensmodel(lensmodel)
        Ni = self.clib.mrcal_lensmodel_num_params(C.byref(m))
        vb, models, which, lead = self._broadcast_models(v, 3, intrinsics_data, Ni, "v")
        N = vb.shape[0]
        q     = np.empty((N,2))
        dq_dv = np.empty((N,2,3))  if get_gradients else None
        dq_di = np.empty((N,2,Ni)) if get_gradients else None
        f = self.clib.mrcal_project
        f.restype  = C.c_bool
        f.argtypes = [C.c_void_p]*4 + [C.c_int, C.c_void_p, C.c_void_p]
        for im in range(models.shape[0]):
            sel = slice(None) if which is None else np.nonzero(which == im)[0]
            vi = np.ascontiguousarray(vb[sel]); n = vi.shape[0]
            if n == 0: continue
            qi = np.empty((n,2)); gv = np.empty((n,2,3)) if get_gradients else None
            gi = np.empty((n,2,Ni)) if get_gradients else None
            intr = np.ascontiguousarray(models[im])
            if not f(_ptr(qi), _ptr(gv) if get_gradients else None, _ptr(gi) if get_gradients else None,
                     _ptr(vi), n, C.byref(m), _ptr(intr)):
                raise RuntimeError("mrcal_project() failed!" + self._last_error())
            q[sel] = qi
            if get_gradients: dq_dv[sel] = gv; dq_di[sel] = gi
        q = q.reshape(lead + (2,))
        if not get_gradients:
            if out is not None: out[...] = q; return out
            return q
        res = (q, dq_dv.reshape(lead + (2,3)), dq_di.reshape(lead + (2,Ni)))
        if out is not None:
            for o, r in zip(out, res): o[...] = r
            return out
        return res

    def unproject(self, q, lensmodel, intrinsics_data, normalize=False, get_gradients=False, out=None):
        """v = unproject(q): pixel coordinates (...,2) to observation vectors
        (...,3) in camera coordinates, on the GPU (mrcal.unproject(),
        mrcal/projections.py:112-395, over mrcal_unproject(), mrcal.h:401-411).
        Broadcasts over q and over intrinsics_data (...,Nintrinsics). Not
        normalized unless asked. With get_gradients: (v, dv_dq (...,3,2),
        dv_dintrinsics (...,3,Nintrinsics)), derived from the gradients of
        project() at the solution as the reference does; like the reference's,
        the length of v then differs from the no-gradients call unless
        normalize=True"""
        m = self.lib.lensmodel(lensmodel)
        Ni = self.clib.mrcal_lensmodel_num_params(C.byref(m))
        if not self.lib.has_symbol("mrcal_amd_unproject"):
            # a library with the reference's entry points only (the tests' checker): its mrcal_unproject()
            if get_gradients:
                raise RuntimeError("this library has no unproject() gradients")
            qq = np.ascontiguousarray(q, dtype=np.float64)
            intr = np.ascontiguousarray(intrinsics_data, dtype=np.float64)
            vv = np.empty(qq.shape[:-1] + (3,))
            g = self.clib.mrcal_unproject
            g.restype  = C.c_bool
            g.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            if not g(_ptr(vv), _ptr(qq), qq.size // 2, C.byref(m), _ptr(intr)):
                raise RuntimeError("mrcal_unproject() failed!" + self._last_error())
            if normalize: vv /= np.linalg.norm(vv, axis=-1, keepdims=True)
            return vv
        qb, models, which, lead = self._broadcast_models(q, 2, intrinsics_data, Ni, "q")
        N = qb.shape[0]
        v     = np.empty((N,3))
        dv_dq = np.empty((N,3,2))  if get_gradients else None
        dv_di = np.empty((N,3,Ni)) if get_gradients else None
        f = self.clib.mrcal_amd_unproject
        f.restype  = C.c_bool
        f.argtypes = [C.c_void_p]*4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_bool]
        for im in range(models.shape[0]):
            sel = slice(None) if which is None else np.nonzero(which == im)[0]
            qi = np.ascontiguousarray(qb[sel]); n = qi.shape[0]
            if n == 0: continue
            vi = np.empty((n,3)); gq = np.empty((n,3,2)) if get_gradients else None
            gi = np.empty((n,3,Ni)) if get_gradients else None
            intr = np.ascontiguousarray(models[im])
            if not f(_ptr(vi), _ptr(gq) if get_gradients else None, _ptr(gi) if get_gradients else None,
                     _ptr(qi), n, C.byref(m), _ptr(intr), bool(normalize)):
                raise RuntimeError("mrcal_unproject() failed!" + self._last_error())
            v[sel] = vi
            if get_gradients: dv_dq[sel] = gq; dv_di[sel] = gi
        v = v.reshape(lead + (3,))
        if not get_gradients:
            if out is not None: out[...] = v; return out
            return v
        res = (v, dv_dq.reshape(lead + (3,2)), dv_di.reshape(lead + (3,Ni)))
        if out is not None:
            for o, r in zip(out, res): o[...] = r
            return out
        return res

    def _last_error(self):
        if self.lib.has_symbol("mrcal_amd_last_error"):
            f = self.clib.mrcal_amd_last_error
            f.restype = C.c_char_p
            s = f()
            if s:
                return " " + s.decode()
        return ""

    # ------------------------------------------------------------------ #
    # layout helpers (mrcal-pywrap.c:2178-3594)                           #
    # ------------------------------------------------------------------ #
    def _layout_args(self, kwargs, need_lensmodel=True):
        """Counts come either explicitly (Ncameras_intrinsics=...) or from the
        arrays; explicit wins; absent means 0 (mrcal-pywrap.c:2318-2326)"""
        kw = dict(kwargs)
        def count(explicit, *arrays):
            v = kw.get(explicit, None)
            if v is not None and v >= 0:
                return int(v)
            for a in arrays:
                arr = kw.get(a, None)
                if arr is not None and not isinstance(arr, str):
                    return int(np.asarray(arr).shape[0])
            return 0
        Nci  = count("Ncameras_intrinsics", "intrinsics")
        Nce  = count("Ncameras_extrinsics", "rt_cam_ref", "extrinsics_rt_fromref")
        Nf   = count("Nframes",             "rt_ref_frame", "frames_rt_toref")
        Np   = count("Npoints",             "points")
        Nob  = count("Nobservations_board", "observations_board")
        Nop  = count("Nobservations_point", "observations_point")
        Npf  = int(kw.get("Npoints_fixed", 0) or 0)
        ob   = kw.get("observations_board", None)
        height_n = kw.get("calibration_object_height_n", None)
        width_n  = kw.get("calibration_object_width_n",  None)
        if ob is not None and Nob > 0 and np.asarray(ob).ndim == 4:
            if height_n is None: height_n = ob.shape[1]
            if width_n  is None: width_n  = ob.shape[2]
        if height_n is None: height_n = 0
        if width_n  is None: width_n  = 0

        def flag(name, default):
            v = kw.get(name, None)
            if v is None or (isinstance(v, (int, np.integer)) and v < 0):
                return bool(default)
            return bool(v)
        sel = ProblemSelections.make(
            do_optimize_intrinsics_core        = flag("do_optimize_intrinsics_core",        Nci > 0),
            do_optimize_intrinsics_distortions = flag("do_optimize_intrinsics_distortions", Nci > 0),
            do_optimize_extrinsics             = flag("do_optimize_extrinsics",             Nce > 0),
            do_optimize_frames                 = flag("do_optimize_frames",                 Nf  > 0),
            do_optimize_calobject_warp         = flag("do_optimize_calobject_warp",         Nob > 0),
            do_apply_regularization            = flag("do_apply_regularization",            True),
            do_apply_outlier_rejection         = flag("do_apply_outlier_rejection",         True),
            do_apply_regularization_unity_cam01= flag("do_apply_regularization_unity_cam01",False))
        lensmodel = None
        if need_lensmodel:
            name = kw.get("lensmodel", None)
            if name is None:
                raise RuntimeError("The 'lensmodel' argument is required")
            lensmodel = self.lib.lensmodel(name)
        # the triangulated observations matter to the layout through their set
        # structure only (mrcal-pywrap.c:2278-2300)
        tri, Ntri = None, 0
        tri_idx = kw.get("indices_point_triangulated_camintrinsics_camextrinsics", None)
        if tri_idx is not None and np.asarray(tri_idx).shape[0] > 0:
            tri_idx = np.ascontiguousarray(tri_idx, dtype=np.int32)
            c_tri = self._fill_triangulated(np.ones((tri_idx.shape[0],3)), tri_idx, None, None)
            tri, Ntri = c_tri, tri_idx.shape[0]
        return dict(Nci=Nci, Nce=Nce, Nf=Nf, Np=Np, Npf=Npf, Nob=Nob, Nop=Nop,
                    width_n=int(width_n), height_n=int(height_n), sel=sel, lensmodel=lensmodel,
                    tri=tri, Ntri=Ntri)

    @staticmethod
    def _none_if_negative(i):
        return None if i < 0 else i

    def _state(self, a):
        return (a["Nci"], a["Nce"], a["Nf"], a["Np"], a["Npf"], a["Nob"], a["sel"],
                C.byref(a["lensmodel"]))

    def state_index_intrinsics(self, icam_intrinsics, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_state_index_intrinsics(icam_intrinsics, *self._state(a)))
    def state_index_extrinsics(self, icam_extrinsics, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_state_index_extrinsics(icam_extrinsics, *self._state(a)))
    def state_index_frames(self, iframe, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_state_index_frames(iframe, *self._state(a)))
    def state_index_points(self, i_point, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_state_index_points(i_point, *self._state(a)))
    def state_index_calobject_warp(self, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_state_index_calobject_warp(*self._state(a)))

    def num_states(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_states(*self._state(a))
    def num_states_intrinsics(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_states_intrinsics(a["Nci"], a["sel"], C.byref(a["lensmodel"]))
    def num_states_extrinsics(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_states_extrinsics(a["Nce"], a["sel"])
    def num_states_frames(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_states_frames(a["Nf"], a["sel"])
    def num_states_points(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_states_points(a["Np"], a["Npf"], a["sel"])
    def num_states_calobject_warp(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_states_calobject_warp(a["sel"], a["Nob"])
    def num_intrinsics_optimization_params(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_intrinsics_optimization_params(a["sel"], C.byref(a["lensmodel"]))

    def measurement_index_boards(self, i_observation_board, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_measurement_index_boards(
            i_observation_board, a["Nob"], a["Nop"], a["width_n"], a["height_n"]))
    def num_measurements_boards(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_measurements_boards(a["Nob"], a["width_n"], a["height_n"])
    def measurement_index_points(self, i_observation_point, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_measurement_index_points(
            i_observation_point, a["Nob"], a["Nop"], a["width_n"], a["height_n"]))
    def num_measurements_points(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_measurements_points(a["Nop"])
    def measurement_index_points_triangulated(self, i_point_triangulated=0, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_measurement_index_points_triangulated(
            i_point_triangulated, a["Nob"], a["Nop"], _ptr(a["tri"]), a["Ntri"], a["width_n"], a["height_n"]))
    def decode_observation_indices_points_triangulated(self, imeasurement, **kw):
        """mrcal-pywrap.c:3336-3420: which pair of triangulated observations a
        measurement (counted from the first triangulated one) belongs to"""
        a = self._layout_args(kw, need_lensmodel=False)
        if a["Ntri"] <= 0:
            raise RuntimeError("No triangulated points in this solve. Nothing to decode")
        out = [C.c_int(0) for _ in range(6)]
        if not self.clib.mrcal_decode_observation_indices_points_triangulated(
                *[C.byref(v) for v in out], int(imeasurement), _ptr(a["tri"]), a["Ntri"]):
            raise RuntimeError("Error decoding indices")
        names = ("iobservation0", "iobservation1", "iobservation_point0",
                 "Nobservations_this_point", "Nmeasurements_this_point", "ipoint")
        return {k: v.value for k, v in zip(names, out)}
    def num_measurements_points_triangulated(self, **kw):
        a = self._layout_args(kw, need_lensmodel=False)
        return self.clib.mrcal_num_measurements_points_triangulated(_ptr(a["tri"]), a["Ntri"])
    def measurement_index_regularization(self, **kw):
        a = self._layout_args(kw)
        return self._none_if_negative(self.clib.mrcal_measurement_index_regularization(
            _ptr(a["tri"]), a["Ntri"], a["width_n"], a["height_n"],
            a["Nci"], a["Nce"], a["Nf"], a["Np"], a["Npf"], a["Nob"], a["Nop"],
            a["sel"], C.byref(a["lensmodel"])))
    def num_measurements_regularization(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_measurements_regularization(*self._state(a))
    def num_measurements(self, **kw):
        a = self._layout_args(kw)
        return self.clib.mrcal_num_measurements(
            a["Nob"], a["Nop"], _ptr(a["tri"]), a["Ntri"], a["width_n"], a["height_n"],
            a["Nci"], a["Nce"], a["Nf"], a["Np"], a["Npf"], a["sel"], C.byref(a["lensmodel"]))

    def corresponding_icam_extrinsics(self, icam_intrinsics, **kw):
        a = self._layout_args(kw, need_lensmodel=False)
        idx_board = kw.get("indices_frame_camintrinsics_camextrinsics", None)
        idx_point = kw.get("indices_point_camintrinsics_camextrinsics", None)
        idx_board = np.zeros((0,3), np.int32) if idx_board is None else idx_board
        idx_point = np.zeros((0,3), np.int32) if idx_point is None else idx_point
        c_board = np.empty((idx_board.shape[0],), dtype=observation_board_dtype)
        c_board["iframe"], c_board["icam_intrinsics"], c_board["icam_extrinsics"] = idx_board.T
        c_point = np.empty((idx_point.shape[0],), dtype=observation_point_dtype)
        c_point["i_point"], c_point["icam_intrinsics"], c_point["icam_extrinsics"] = idx_point.T
        if not (0 <= icam_intrinsics < a["Nci"]):
            raise RuntimeError(f"The given icam_intrinsics={icam_intrinsics} is out of bounds. Must be >= 0 and < {a['Nci']}")
        out = C.c_int(-100)
        if not self.clib.mrcal_corresponding_icam_extrinsics(
                C.byref(out), icam_intrinsics, a["Nci"], a["Nce"],
                c_board.shape[0], _ptr(c_board), c_point.shape[0], _ptr(c_point)):
            raise RuntimeError("Error calling mrcal_corresponding_icam_extrinsics()" + self._last_error())
        return out.value

    def _pack_unpack(self, f, b, kw):
        a = self._layout_args(kw)
        if not isinstance(b, np.ndarray) or b.dtype != np.float64 or not b.flags["C_CONTIGUOUS"]:
            raise RuntimeError("The given array MUST be a C-contiguous numpy array of float64")
        Nstate = self.clib.mrcal_num_states(*self._state(a))
        if b.ndim < 1 or b.shape[-1] != Nstate:
            raise RuntimeError(f"The given array MUST have shape (...,{Nstate}); got {b.shape}")
        rows = b.reshape(-1, Nstate)
        for i in range(rows.shape[0]):
            f(rows[i].ctypes.data_as(_cabi.c_double_p), *self._state(a))
        return None

    def pack_state(self, b, **kw):
        """in place: unpacked units -> the unitless state the solver sees"""
        return self._pack_unpack(self.clib.mrcal_pack_solver_state_vector, b, kw)
    def unpack_state(self, b, **kw):
        """in place: unitless solver state -> unpacked units"""
        return self._pack_unpack(self.clib.mrcal_unpack_solver_state_vector, b, kw)

    def lensmodel_num_params(self, lensmodel):
        m = self.lib.lensmodel(lensmodel)
        return self.clib.mrcal_lensmodel_num_params(C.byref(m))
