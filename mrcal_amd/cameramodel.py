"""The .cameramodel file format and the (de)serialisation of optimization_inputs:
what lets a stored calibration be replayed through this solver.

Mirrors the interface of the reference's mrcal.cameramodel class
(mrcal/cameramodel.py:390-2077) for the native format:

    m = cameramodel('camera0.cameramodel')          # file name, open file, or another cameramodel
    m = cameramodel(intrinsics = ('LENSMODEL_OPENCV8', data), imagersize = (w,h), rt_cam_ref = rt)
    m = cameramodel(optimization_inputs = optimization_inputs, icam_intrinsics = 0)
    lensmodel, data = m.intrinsics();  m.rt_cam_ref();  m.imagersize()
    stats = mrcal_amd.optimize(**m.optimization_inputs())       # re-run the solve the model came from
    m.write('out.cameramodel', note = "...")

The file is a python dict literal (keys lensmodel, intrinsics, rt_cam_ref [and
its pre-2.5 name extrinsics], imagersize, optionally valid_intrinsics_region,
icam_intrinsics, icam_extrinsics, optimization_inputs). optimization_inputs is
stored as base-85 text of a numpy .npz (compressed, pickling disabled), None
stored as '' (cameramodel.py:160-360), under the PRE-2.5 key names
(frames_rt_toref, extrinsics_rt_fromref) so that files stay readable by old
tools; reading accepts both generations of names.

Not provided: the .cahvor and OpenCV/ROS yaml formats the reference also reads
(other projects' formats; out of this path's scope).
"""
import ast
import base64
import io
import numbers
import warnings

import numpy as np

from . import poseutils as _pu


class CameramodelParseException(Exception):
    """the text is not a .cameramodel"""


_RENAMED_INPUTS = (("frames_rt_toref", "rt_ref_frame"),
                   ("extrinsics_rt_fromref", "rt_cam_ref"))
# recent additions whose default value is left out of the file (older readers do not know them)
_OMIT_WHEN_DEFAULT = ("do_apply_regularization_unity_cam01",
                      "observations_point_triangulated",
                      "indices_point_triangulated_camintrinsics_camextrinsics")


def _is_poison(v):
    return isinstance(v, str) and v.startswith("ERROR:")


def _serialize_optimization_inputs(optimization_inputs):
    """dict -> base-85 bytes of a compressed .npz (cameramodel.py:160-307)"""
    d = {}
    for k, v in optimization_inputs.items():
        if _is_poison(v): continue                   # the marker a read leaves under an old name
        if v is None: v = ""
        if k in _OMIT_WHEN_DEFAULT:
            if isinstance(v, np.ndarray):
                if v.size == 0: continue
            elif not v:
                continue
        d[k] = v
    # on disk: the old names only
    for old, new in _RENAMED_INPUTS:
        have_old = old in d and not (isinstance(d[old], str) and d[old] == "")
        have_new = new in d and not (isinstance(d[new], str) and d[new] == "")
        if have_old and have_new:
            try:    same = not np.any(np.asarray(d[old]) - np.asarray(d[new]))
            except Exception: same = False
            if not same:
                raise Exception(f"optimization_inputs has both '{old}' and '{new}', and they differ")
        elif have_new:
            d[old] = d[new]
        d.pop(new, None)
    f = io.BytesIO()
    np.savez_compressed(f, **d)
    return base64.b85encode(f.getvalue())


def _deserialize_optimization_inputs(data_bytes):
    """the inverse (cameramodel.py:310-387): scalars come back as python scalars,
    '' as None, old key names as the new ones. The old names stay behind as
    strings that explain the renaming, exactly as the reference leaves them, so
    that a dict read here and handed to code written for either generation
    behaves the same way; optimize() ignores them"""
    try:
        z = np.load(io.BytesIO(base64.b85decode(data_bytes)), allow_pickle=False)
    except Exception as e:
        raise CameramodelParseException(f"could not decode the optimization_inputs: {e}")
    d = {}
    for k in z.keys():
        a = z[k]
        if a.shape == (): a = a.item()
        if isinstance(a, str) and a == "": a = None
        elif isinstance(a, np.ndarray) and not a.dtype.isnative:
            a = a.astype(a.dtype.newbyteorder("="))
        d[k] = a
    for old, new in (("do_optimize_intrinsic_core", "do_optimize_intrinsics_core"),
                     ("do_optimize_intrinsic_distortions", "do_optimize_intrinsics_distortions")):
        if old in d and new not in d: d[new] = d.pop(old)
    for old, new in _RENAMED_INPUTS:
        if old in d and new not in d: d[new] = d[old]
        d[old] = (f'ERROR: mrcal 2.5 renamed optimization_inputs fields: "{old}" -> "{new}". '
                  "Please update your code to use the new name")
    d.pop("calibration_object_width_n", None)
    d.pop("calibration_object_height_n", None)
    if d.get("rt_cam_ref") is None:
        d["rt_cam_ref"] = np.zeros((0, 6))
    return d


def _check_imagersize(imagersize):
    try:    ok = len(imagersize) == 2 and all(s > 0 and s == int(s) for s in imagersize)
    except Exception: ok = False
    if not ok: raise Exception("The imagersize must be an iterable of two positive integers")


def _check_intrinsics(intrinsics):
    from . import lensmodel_num_params
    try:    ok = len(intrinsics) == 2
    except Exception: ok = False
    if not ok: raise Exception("Valid intrinsics are a (lensmodel, intrinsics_data) pair")
    lensmodel, data = intrinsics
    try:    N = len(data)
    except Exception: raise Exception("Valid intrinsics are (lensmodel, intrinsics_data) where intrinsics_data has a length")
    Nwant = lensmodel_num_params(lensmodel)
    if N != Nwant:
        raise Exception(f"Mismatched Nintrinsics. Got {N}, but model {lensmodel} must have {Nwant}")
    for x in data:
        if not isinstance(x, numbers.Number):
            raise Exception(f"All intrinsics elements should be numeric, but '{x}' isn't")


def _check_rt(rt):
    try:    ok = len(rt) == 6 and all(isinstance(x, numbers.Number) for x in rt)
    except Exception: ok = False
    if not ok: raise Exception("Valid extrinsics are an iterable of 6 numbers")


def _check_region(region):
    if region is None: return
    if not (isinstance(region, np.ndarray) and region.ndim == 2 and region.shape[1] == 2 and
            (region.shape[0] >= 4 or region.shape[0] == 0)):
        raise Exception("The valid-intrinsics region must be a numpy array of shape (N,2) with N >= 4 or N == 0")
    if region.size > 0 and np.sum((region[0] - region[-1])**2) > 1e-6:
        raise Exception("The valid-intrinsics region must be a closed contour: first point == last point")


class _Solve:
    """where a model came from: the serialized optimization_inputs (bytes, or None) and which of that solve's
    cameras this is"""
    __slots__ = ("blob", "icam_i", "icam_e")
    def __init__(self, blob=None, icam_i=None, icam_e=None):
        self.blob, self.icam_i, self.icam_e = blob, icam_i, icam_e


class cameramodel:
    """one camera: lens model + intrinsics, imager size, pose relative to the
    reference frame, optionally the solve that produced it"""

    # ------------------------------------------------------------------ construction
    def __init__(self, file_or_model=None, *,
                 intrinsics=None, imagersize=None,
                 Rt_ref_cam=None, Rt_cam_ref=None, rt_ref_cam=None, rt_cam_ref=None,
                 extrinsics_Rt_toref=None, extrinsics_Rt_fromref=None,
                 extrinsics_rt_toref=None, extrinsics_rt_fromref=None,
                 optimization_inputs=None, icam_intrinsics=None, icam_extrinsics=None,
                 valid_intrinsics_region=None):
        poses = dict(Rt_ref_cam=Rt_ref_cam if Rt_ref_cam is not None else extrinsics_Rt_toref,
                     Rt_cam_ref=Rt_cam_ref if Rt_cam_ref is not None else extrinsics_Rt_fromref,
                     rt_ref_cam=rt_ref_cam if rt_ref_cam is not None else extrinsics_rt_toref,
                     rt_cam_ref=rt_cam_ref if rt_cam_ref is not None else extrinsics_rt_fromref)
        Nposes = sum(v is not None for v in (Rt_ref_cam, Rt_cam_ref, rt_ref_cam, rt_cam_ref, extrinsics_Rt_toref,
                                             extrinsics_Rt_fromref, extrinsics_rt_toref, extrinsics_rt_fromref))
        discrete = intrinsics is not None or imagersize is not None or Nposes > 0
        from_inputs = optimization_inputs is not None or icam_intrinsics is not None or icam_extrinsics is not None

        self._valid_intrinsics_region = None
        self._solve = _Solve()

        if file_or_model is not None:
            if discrete or from_inputs:
                raise Exception("'file_or_model' specified, so none of the other inputs should be")
            if isinstance(file_or_model, cameramodel):
                o = file_or_model
                self._imagersize = np.array(o._imagersize, dtype=np.int32)
                self._rt_cam_ref = np.array(o._rt_cam_ref, dtype=float)
                self._intrinsics = (str(o._intrinsics[0]), np.array(o._intrinsics[1], dtype=float))
                if o._valid_intrinsics_region is not None:
                    self._valid_intrinsics_region = np.array(o._valid_intrinsics_region, dtype=float)
                self._solve = _Solve(o._solve.blob, o._solve.icam_i, o._solve.icam_e)
            elif isinstance(file_or_model, str):
                if file_or_model == "-":
                    import sys
                    if sys.stdin.isatty():
                        raise Exception("Trying to read a model from standard input, but nothing is being redirected into it")
                    self._read_into_self(sys.stdin.read(), "STDIN")
                else:
                    with open(file_or_model, "r") as f:
                        self._read_into_self(f.read(), f"file '{file_or_model}'")
            else:
                self._read_into_self(file_or_model.read(), getattr(file_or_model, "name", "file object"))
            return

        if discrete:
            if from_inputs:
                raise Exception("discrete values specified, so none of the other inputs should be")
            if intrinsics is None or imagersize is None or Nposes > 1:
                raise Exception("Discrete values given. Need 'intrinsics' AND 'imagersize' AND optionally ONE of the extrinsics")
            self._rt_cam_ref = np.zeros(6)
            for name, value in poses.items():
                if value is not None: getattr(self, name)(value)
            self.intrinsics(intrinsics, imagersize=imagersize)
        elif from_inputs:
            if optimization_inputs is None:
                raise Exception("icam_intrinsics or icam_extrinsics are given, so optimization_inputs MUST be given")
            if icam_intrinsics is None:
                raise Exception("optimization_inputs given, so icam_intrinsics MUST be given")
            self.intrinsics((optimization_inputs["lensmodel"], optimization_inputs["intrinsics"][icam_intrinsics]),
                            imagersize=optimization_inputs["imagersizes"][icam_intrinsics],
                            optimization_inputs=optimization_inputs,
                            icam_intrinsics=icam_intrinsics, icam_extrinsics=icam_extrinsics)
            if self._solve.icam_e < 0:
                self.rt_cam_ref(np.zeros(6))
            else:
                rt = optimization_inputs.get("rt_cam_ref")
                if rt is None or _is_poison(rt): rt = optimization_inputs["extrinsics_rt_fromref"]
                self.rt_cam_ref(rt[self._solve.icam_e])
        else:
            raise Exception("Need a filename or a cameramodel object or discrete arrays or optimization_inputs")

        if valid_intrinsics_region is not None:
            try:
                self.valid_intrinsics_region(valid_intrinsics_region)
            except Exception as e:
                warnings.warn(f"Invalid valid_intrinsics region; skipping: '{e}'")

    # ------------------------------------------------------------------ text form
    def _read_into_self(self, text, what=None):
        try:
            model = ast.literal_eval(text)
            if not isinstance(model, dict): raise ValueError("not a dict")
        except Exception:
            raise CameramodelParseException("Failed to parse cameramodel" + (f" '{what}'" if what else ""))

        def renamed(old, new, same=False):
            if old not in model: return
            if new not in model:
                model[new] = model.pop(old)
            elif same:
                try:    equal = np.all(np.abs(np.array(model[old]) - np.array(model[new])) < 1e-9)
                except Exception as e:
                    raise CameramodelParseException(f"'{new}' and '{old}' both given; couldn't compare them: {e}")
                if not equal:
                    raise CameramodelParseException(f"'{new}' and '{old}' both given, and they're NOT the same")
        renamed("distortion_model", "lensmodel")
        renamed("lens_model", "lensmodel")
        renamed("icam_intrinsics_optimization_inputs", "icam_intrinsics")
        renamed("extrinsics", "rt_cam_ref", same=True)

        missing = {"lensmodel", "intrinsics", "rt_cam_ref", "imagersize"} - set(model.keys())
        if missing:
            raise CameramodelParseException(f"Model is missing the keys {sorted(missing)}; it has {sorted(model.keys())}")
        if isinstance(model["lensmodel"], bytes): model["lensmodel"] = model["lensmodel"].decode()
        model["lensmodel"] = model["lensmodel"].replace("DISTORTION", "LENSMODEL")

        intrinsics = (model["lensmodel"], np.array(model["intrinsics"], dtype=float))
        _check_imagersize(model["imagersize"])
        _check_intrinsics(intrinsics)
        _check_rt(model["rt_cam_ref"])

        region = None
        if "valid_intrinsics_region" in model:
            region = (np.array(model["valid_intrinsics_region"], dtype=float)
                      if len(model["valid_intrinsics_region"]) > 0 else np.zeros((0, 2)))
            try:
                _check_region(region)
            except Exception as e:
                warnings.warn(f"Invalid valid_intrinsics region; skipping: '{e}'")
                region = None

        self._intrinsics = intrinsics
        self._valid_intrinsics_region = _pu.close_contour(region)
        self._rt_cam_ref = np.array(model["rt_cam_ref"], dtype=float)
        self._imagersize = np.array(model["imagersize"], dtype=np.int32)

        if "optimization_inputs" in model:
            if not isinstance(model["optimization_inputs"], bytes):
                raise CameramodelParseException("'optimization_inputs' is given, but it's not a byte string")
            icam = model.get("icam_intrinsics")
            if not isinstance(icam, int) or icam < 0:
                raise CameramodelParseException("'optimization_inputs' is given, so 'icam_intrinsics' must be an int >= 0")
            icam_e = model.get("icam_extrinsics")
            if icam_e is not None and not isinstance(icam_e, int):
                raise CameramodelParseException("'icam_extrinsics' is given, but it's not an int")
            self._solve.blob = model["optimization_inputs"]
            self._solve.icam_i = icam
            self._solve.icam_e = icam_e
        else:
            if "icam_intrinsics" in model or "icam_extrinsics" in model:
                raise CameramodelParseException("'optimization_inputs' is NOT given, but icam_intrinsics or icam_extrinsics ARE")
            self._solve.blob = None
            self._solve.icam_i = None
            self._solve.icam_e = None

    def _entries(self):
        """the file as data: (key, python literal text, comment lines above it, trailing comment, blank line after).
        The byte layout is the format's (mrcal/cameramodel.py:503-558 writes the same bytes; files are compared
        byte for byte in tests/test_cameramodel.py)"""
        def row(v): return "[" + "".join(f" {x:.10g}," for x in v) + "]"
        lens, data = self._intrinsics
        out = [("lensmodel", f" '{lens}'", (), None, True),
               ("intrinsics", " " + row(data), ("intrinsics are fx,fy,cx,cy,distortion0,distortion1,....",), None, True)]
        if self._valid_intrinsics_region is not None:
            body = "".join(f"    [ {q[0]:.10g}, {q[1]:.10g} ],\n" for q in self._valid_intrinsics_region)
            out.append(("valid_intrinsics_region", " [\n" + body + "]", (), None, True))
        out.append(("rt_cam_ref", " " + row(self._rt_cam_ref), (), None, False))
        out.append(("extrinsics", " " + row(self._rt_cam_ref), (), "for compatibility with mrcal < 2.5", True))
        out.append(("imagersize", f" [ {int(self._imagersize[0])}, {int(self._imagersize[1])},]", (), None, True))
        for key, v in (("icam_intrinsics", self._solve.icam_i), ("icam_extrinsics", self._solve.icam_e)):
            if v is not None: out.append((key, f" {v:d}", (), None, False))
        out.append((None, None, (), None, True))
        if self._solve.blob is not None:
            out.append(("optimization_inputs", f" {self._solve.blob}",
                        ("Everything the solve that produced this model was given (ALL the observations of",
                         "ALL its cameras), at its optimum: numpy .npz, compressed, base-85. Needed for",
                         "projection uncertainties and for re-running the solve. Editing the intrinsics",
                         "invalidates it; moving the camera (the extrinsics) does not"), None, True))
        return out

    def _write(self, f, note=None):
        text = ["# " + line + "\n" for line in (note.splitlines() if note is not None else ())]
        text.append("{\n")
        pad = {"lensmodel": " "}       # (the format aligns this one value with the next)
        for key, literal, above, trailing, blank in self._entries():
            text.extend(f"    # {c}\n" for c in above)
            if key is not None:
                text.append(f"    '{key}':{pad.get(key, '')}{literal},{' # ' + trailing if trailing else ''}\n")
            if blank: text.append("\n")
        text.append("}\n")
        f.write("".join(text))

    def write(self, f, *, note=None, cahvor=False, opencv=False):
        """to a file name or an open text file"""
        if cahvor or opencv:
            raise NotImplementedError("only the native .cameramodel format is written")
        if isinstance(f, str):
            with open(f, "w") as fp: self._write(fp, note)
        else:
            self._write(f, note)

    def __str__(self):
        f = io.StringIO()
        self._write(f)
        return f.getvalue()

    def __repr__(self):
        return f"mrcal_amd.cameramodel({self._intrinsics[0]}, imagersize={tuple(int(x) for x in self._imagersize)})"

    # ------------------------------------------------------------------ components
    def intrinsics(self, intrinsics=None, *, imagersize=None, optimization_inputs=None,
                   icam_intrinsics=None, icam_extrinsics=None):
        """getter without arguments: (lensmodel, intrinsics_data). Setter: new
        intrinsics and, together with them or not at all, the solve they came
        from; the valid-intrinsics region is dropped"""
        if intrinsics is None and imagersize is None and optimization_inputs is None and icam_intrinsics is None:
            return (str(self._intrinsics[0]), np.array(self._intrinsics[1], dtype=float))
        if imagersize is None: imagersize = self._imagersize
        _check_imagersize(imagersize)
        _check_intrinsics(intrinsics)
        if optimization_inputs is not None:
            if not isinstance(optimization_inputs, dict):
                raise Exception(f"'optimization_inputs' must be a dict. Instead got {type(optimization_inputs)}")
            if not isinstance(icam_intrinsics, (int, np.integer)) or icam_intrinsics < 0:
                raise Exception("optimization_inputs is given, so icam_intrinsics must be an int >= 0")
        elif icam_intrinsics is not None or icam_extrinsics is not None:
            raise Exception("icam_intrinsics and icam_extrinsics make sense ONLY together with optimization_inputs")
        self._imagersize = np.array(imagersize, dtype=np.int32)
        self._intrinsics = (str(intrinsics[0]), np.array(intrinsics[1], dtype=float))
        if optimization_inputs is not None:
            self._solve.blob = _serialize_optimization_inputs(optimization_inputs)
            self._solve.icam_i = int(icam_intrinsics)
            if icam_extrinsics is None:
                from . import corresponding_icam_extrinsics
                clean = {k: v for k, v in optimization_inputs.items() if not _is_poison(v)}
                try:
                    icam_extrinsics = corresponding_icam_extrinsics(int(icam_intrinsics), **clean)
                except Exception as e:
                    raise Exception("optimization_inputs given, but icam_extrinsics not given, and it cannot be "
                                    f"inferred; are the cameras moving? Error: {e}")
            self._solve.icam_e = int(icam_extrinsics)
        else:
            self._solve.blob = None
            self._solve.icam_i = None
            self._solve.icam_e = None
        self._valid_intrinsics_region = None

    def _pose(self, value, to_stored, from_stored):
        if value is None: return from_stored(self._rt_cam_ref)
        self._rt_cam_ref = np.array(to_stored(np.asarray(value, dtype=float)), dtype=float)
        return True

    def rt_cam_ref(self, rt=None):
        """reference -> camera, (6,): getter, or setter when given"""
        return self._pose(rt, lambda v: v, lambda s: np.array(s))
    def rt_ref_cam(self, rt=None):
        return self._pose(rt, _pu.invert_rt, _pu.invert_rt)
    def Rt_cam_ref(self, Rt=None):
        return self._pose(Rt, _pu.rt_from_Rt, _pu.Rt_from_rt)
    def Rt_ref_cam(self, Rt=None):
        return self._pose(Rt, lambda v: _pu.rt_from_Rt(_pu.invert_Rt(v)), lambda s: _pu.invert_Rt(_pu.Rt_from_rt(s)))
    # pre-2.5 names
    extrinsics_rt_fromref = rt_cam_ref
    extrinsics_rt_toref   = rt_ref_cam
    extrinsics_Rt_fromref = Rt_cam_ref
    extrinsics_Rt_toref   = Rt_ref_cam

    def imagersize(self, *args, **kwargs):
        if args or kwargs:
            raise Exception("imagersize() is NOT a setter. Please use intrinsics() to set them all together")
        return np.array(self._imagersize, dtype=np.int32)

    def valid_intrinsics_region(self, valid_intrinsics_region=None):
        """closed contour (N,2) in pixels, None for 'not defined', (0,2) for 'valid nowhere'"""
        if valid_intrinsics_region is None:
            return None if self._valid_intrinsics_region is None else np.array(self._valid_intrinsics_region)
        region = _pu.close_contour(np.asarray(valid_intrinsics_region, dtype=float))
        _check_region(region)
        self._valid_intrinsics_region = region
        return True

    def valid_intrinsics_region_reset(self):
        self._valid_intrinsics_region = None

    def optimization_inputs(self):
        """the kwargs of optimize()/optimizer_callback() at the optimum this model
        came from, or None"""
        if self._solve.blob is None: return None
        d = _deserialize_optimization_inputs(self._solve.blob)
        d["verbose"] = False
        return d

    def optimization_inputs_reset(self):
        self._solve.blob = None

    def icam_intrinsics(self):
        return self._solve.icam_i

    def icam_extrinsics(self):
        """which camera pose of the solve this is (<0: the reference), or None
        without optimization_inputs. Files older than the key: inferred from the
        stored solve, which works for stationary cameras"""
        if self._solve.icam_i is None: return None
        if self._solve.icam_e is None:
            from . import corresponding_icam_extrinsics
            d = {k: v for k, v in self.optimization_inputs().items() if not _is_poison(v)}
            self._solve.icam_e = int(corresponding_icam_extrinsics(self._solve.icam_i, **d))
        return self._solve.icam_e

    def _optimization_inputs_match(self, other):
        return self._solve.blob == other._solve.blob

    def _extrinsics_moved_since_calibration(self):
        icam = self.icam_extrinsics()
        if icam < 0: return np.max(np.abs(self._rt_cam_ref)) > 0.0
        return np.max(np.abs(self._rt_cam_ref - self.optimization_inputs()["rt_cam_ref"][icam])) > 1e-6
