"""the .cameramodel format and the optimization_inputs
(de)serialisation (reference: mrcal/cameramodel.py:160-360, 503-690; the C
reader mrcal.h:858-890, cameramodel-parser.re), and what they are for: replaying
a stored REAL calibration through the GPU path.

CPU:
  - the C reader against the known-answer cases of the reference's own
    test/test-parser-cameramodel.c (accepts: spacing, quote styles, b'' keys,
    tuples, comments everywhere, unknown keys with nested values, both names of
    the extrinsics; rejects: trailing garbage, duplicate keys, missing keys,
    wrong intrinsics counts, disagreeing extrinsics), the _into variants'
    buffer-size protocol, unterminated buffers, write -> read
  - the python class: discrete construction, all four pose views, text round
    trip, legacy key names, validation errors, optimization_inputs round trip
    (None <-> '', scalars, bool, old names on disk / new names in memory)
  - the committed real calibrations parse, and the reference's own code
    (oracle/_ref) accepts what was deserialised
GPU:
  - x and J of the real calibration at its stored optimum == the reference's
  - optimize() from the stored optimum stays there; from a perturbed state it
    comes back to it, with the reference's outlier count
"""
import ctypes as C
import io
import os
import numpy as np
import pytest

from conftest import ROOT, GOLDEN_DIR, relative_error


# --------------------------------------------------------------------- the C reader
class LensModel(C.Structure):
    _fields_ = [("type", C.c_int), ("config", C.c_double)]            # 16 bytes, the union at offset 8
class CameraModelHeader(C.Structure):
    _fields_ = [("rt_cam_ref", C.c_double*6), ("imagersize", C.c_uint*2), ("lensmodel", LensModel)]


@pytest.fixture(scope="module")
def clib():
    lib = C.CDLL(os.path.join(ROOT, "mrcal_amd", "libmrcal_amd.so"))
    lib.mrcal_read_cameramodel_string.restype  = C.c_void_p
    lib.mrcal_read_cameramodel_string.argtypes = [C.c_char_p, C.c_int]
    lib.mrcal_read_cameramodel_file.restype    = C.c_void_p
    lib.mrcal_read_cameramodel_file.argtypes   = [C.c_char_p]
    lib.mrcal_free_cameramodel.argtypes        = [C.POINTER(C.c_void_p)]
    lib.mrcal_read_cameramodel_string_into.restype  = C.c_bool
    lib.mrcal_read_cameramodel_string_into.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_int]
    lib.mrcal_read_cameramodel_file_into.restype    = C.c_bool
    lib.mrcal_read_cameramodel_file_into.argtypes   = [C.c_void_p, C.POINTER(C.c_int), C.c_char_p]
    lib.mrcal_write_cameramodel_file.restype   = C.c_bool
    lib.mrcal_write_cameramodel_file.argtypes  = [C.c_char_p, C.c_void_p]
    lib.mrcal_lensmodel_name.restype  = C.c_bool
    lib.mrcal_lensmodel_name.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    return lib


def unpack(lib, p, N):
    h = CameraModelHeader.from_address(p)
    name = C.create_string_buffer(256)
    assert lib.mrcal_lensmodel_name(name, 256, C.addressof(h.lensmodel))
    intr = np.ctypeslib.as_array((C.c_double*N).from_address(p + C.sizeof(CameraModelHeader))).copy()
    return dict(lensmodel=name.value.decode(), rt=np.array(h.rt_cam_ref), size=tuple(h.imagersize), intrinsics=intr)


def read_string(lib, text, length=0):
    b = text if isinstance(text, bytes) else text.encode()
    p = lib.mrcal_read_cameramodel_string(b, length)
    if not p: return None
    out = unpack(lib, p, 12)
    pp = C.c_void_p(p)
    lib.mrcal_free_cameramodel(C.byref(pp))
    assert pp.value is None
    return out


REF = dict(lensmodel="LENSMODEL_CAHVORE_linearity=0.34", rt=np.array((0, 1, 2, 33, 44e4, -55.3e-3)), size=(110, 400),
           intrinsics=np.array((4, 3, 4, 5, 0, 1, 3, 5, 4, 10, 11, 12.)))
L  = "    'lensmodel':  \"LENSMODEL_CAHVORE_linearity=0.34\",\n"
E  = "    'extrinsics': [ 0., 1, 2, 33, 44e4, -55.3E-3, ],\n"
I  = "    'intrinsics': [ 4, 3, 4, 5, 0, 1, 3, 5, 4, 10, 11, 12 ],\n"
S  = "    'imagersize': [110, 400],\n"

# the accept / reject cases of test/test-parser-cameramodel.c:176-330
ACCEPT = {
    "baseline": "{\n" + L + E + I + S + "}\n",
    "spacing, quotes, b'' key, tuple":
        "{\n    'lensmodel' :  b'LENSMODEL_CAHVORE_linearity=0.34',\n    b'extrinsics' :[ 0., 1, 2, 33, 44e4, -55.3E-3, ],\n"
        "    \"intrinsics\": (4, 3, 4, 5, 0, 1, 3, 5, 4, 10, 11, 12 ),    'imagersize': [110, 400],\n\n}\n",
    "comments everywhere":
        " # f {\n#{ 'lensmodel': 'rrr'\n{'lensmodel':  #\"LENSMODEL_CAHVOR\",\n\"LENSMODEL_CAHVORE_linearity=0.34\",\n"
        "    'extrinsics': [ 0., 1, 2, 33, # 44e4, -55.3E-3,\n44e4, -55.3E-3\n#,\n,\n#]\n"
        "],'intrinsics': [ 4, 3, 4,\n5,    0,  \n\n  1, 3, 5, 4, 10, 11, 12 ],\n    'imagersize': [110, 400]\n# }\n}  \n # }\n",
    "unknown keys with strings and nested lists":
        "{\n    'lensmodel':  \"LENSMODEL_CAHVORE_linearity=0.34\", 'f': 5,\n" + E.rstrip("\n") + " 'xxx':\n # fff\n"
        " b'rr','qq': b'asdf;lkj&*()DSFEWR]]{}}}',\n'vvvv': [ 1,2, [4,5],[3,[4,3,[]],444], ]," + I + S + "}\n",
    "rt_cam_ref instead of extrinsics": "{\n" + L + E.replace("extrinsics", "rt_cam_ref") + I + S + "}\n",
    "extrinsics twice, identical": "{\n" + L + E + E + I + S + "}\n",
    "both names, identical": "{\n" + L + E + E.replace("extrinsics", "rt_cam_ref") + I + S + "}\n",
}
REJECT = {
    "trailing garbage": "{\n" + L + E + I + S + "} f\n",
    "lensmodel twice": "{\n" + L + L + E + I + S + "}\n",
    "extrinsics twice, different": "{\n" + L + E + E.replace("55.3", "55.4") + I + S + "}\n",
    "intrinsics twice": "{\n" + L + E + I + I + S + "}\n",
    "imagersize twice": "{\n" + L + E + I + S + S + "}\n",
    "no lensmodel": "{\n" + E + I + S + "}\n",
    "no extrinsics": "{\n" + L + I + S + "}\n",
    "no intrinsics": "{\n" + L + E + S + "}\n",
    "no imagersize": "{\n" + L + E + I + "}\n",
    "too few intrinsics": "{\n" + L + E + I.replace(" 12 ]", "]") + S + "}\n",
    "too many intrinsics": "{\n" + L + E + I.replace("12 ]", "99,88]") + S + "}\n",
    "intrinsics before lensmodel": "{\n" + I + L + E + S + "}\n",
    "no comma between pairs": "{\n" + L + E + I.rstrip(",\n") + "\n" + S + "}\n",
    "no opening brace": L + E + I + S + "}\n",
    "unknown lens model": "{\n" + L.replace("CAHVORE_linearity=0.34", "FISHEYE9") + E + I + S + "}\n",
}


@pytest.mark.parametrize("name", sorted(ACCEPT))
def test_c_reader_accepts(clib, name):
    m = read_string(clib, ACCEPT[name])
    assert m is not None, name
    assert m["lensmodel"] == REF["lensmodel"] and m["size"] == REF["size"]
    assert np.array_equal(m["rt"], REF["rt"]) and np.array_equal(m["intrinsics"], REF["intrinsics"])


@pytest.mark.parametrize("name", sorted(REJECT))
def test_c_reader_rejects(clib, name, capfd):
    assert read_string(clib, REJECT[name]) is None, name
    capfd.readouterr()


def test_c_reader_unterminated_buffer_and_into(clib, tmp_path):
    text = ACCEPT["baseline"].encode()
    # len > 0: the byte behind the text is not a terminator
    m = read_string(clib, text + b"5", len(text))
    assert m is not None and np.array_equal(m["intrinsics"], REF["intrinsics"])
    # caller's buffer: big enough, then one too small -> told how many are needed
    buf = C.create_string_buffer(C.sizeof(CameraModelHeader) + 12*8)
    n = C.c_int(12)
    assert clib.mrcal_read_cameramodel_string_into(buf, C.byref(n), text, 0)
    assert np.array_equal(unpack(clib, C.addressof(buf), 12)["intrinsics"], REF["intrinsics"])
    n = C.c_int(11)
    assert not clib.mrcal_read_cameramodel_string_into(buf, C.byref(n), text, 0)
    assert n.value == 12
    # any other failure: <= 0
    n = C.c_int(12)
    assert not clib.mrcal_read_cameramodel_string_into(buf, C.byref(n), REJECT["no imagersize"].encode(), 0)
    assert n.value <= 0
    # files: write -> read, full precision
    path = str(tmp_path / "m.cameramodel").encode()
    h = CameraModelHeader.from_buffer(buf)
    vals = np.array((1761.181055123456, 1761.25, 1965.7, 1087.5, -1.266096516e-7, 3.59e-9, -2.5e-11, 5.2e-4, 0.0196, 0.0148, -0.0562, 0.05))
    C.memmove(C.addressof(buf) + C.sizeof(CameraModelHeader), vals.ctypes.data, 96)
    assert clib.mrcal_write_cameramodel_file(path, buf)
    p = clib.mrcal_read_cameramodel_file(path)
    assert p
    m = unpack(clib, p, 12)
    assert np.array_equal(m["intrinsics"], vals) and np.array_equal(m["rt"], REF["rt"])
    buf2 = C.create_string_buffer(C.sizeof(CameraModelHeader) + 12*8)
    n = C.c_int(12)
    assert clib.mrcal_read_cameramodel_file_into(buf2, C.byref(n), path)
    assert np.array_equal(unpack(clib, C.addressof(buf2), 12)["intrinsics"], vals)
    assert not clib.mrcal_read_cameramodel_file(b"/nonexistent/x.cameramodel")
    # the python writer's output is the C reader's input
    from mrcal_amd.cameramodel import cameramodel
    pm = cameramodel(intrinsics=("LENSMODEL_OPENCV8", vals), imagersize=(4000, 2200), rt_cam_ref=np.arange(6.)*0.1,
                     valid_intrinsics_region=np.array(((0, 0), (10, 0), (10, 10), (0, 10.))))
    m = read_string(clib, str(pm))
    assert m["lensmodel"] == "LENSMODEL_OPENCV8" and m["size"] == (4000, 2200)
    assert np.allclose(m["intrinsics"], vals, rtol=1e-9, atol=0) and np.allclose(m["rt"], np.arange(6.)*0.1)


# --------------------------------------------------------------------- the python class
def test_python_class_discrete_and_text_round_trip(amd, tmp_path):
    from mrcal_amd.cameramodel import cameramodel, CameramodelParseException
    from mrcal_amd import poseutils as pu
    intr = np.array((1761.18, 1761.25, 1965.7, 1087.5, -0.0127, 0.0359, -0.00025, 0.00053, 0.0197, 0.0148, -0.0562, 0.0500))
    rt = np.array((2e-2, -3e-1, -1e-2, 1., 2, -3.))
    m = cameramodel(intrinsics=("LENSMODEL_OPENCV8", intr), imagersize=(4000, 2200), rt_cam_ref=rt)
    assert m.intrinsics()[0] == "LENSMODEL_OPENCV8" and np.array_equal(m.intrinsics()[1], intr)
    assert np.array_equal(m.imagersize(), (4000, 2200)) and m.optimization_inputs() is None and m.icam_intrinsics() is None
    # the four views of the pose agree
    x = np.array((0.3, -0.2, 5.))
    xc = pu.transform_point_rt(m.rt_cam_ref(), x)
    assert np.allclose(pu.transform_point_Rt(m.Rt_cam_ref(), x), xc)
    assert np.allclose(pu.transform_point_rt(m.rt_ref_cam(), xc), x)
    assert np.allclose(pu.transform_point_Rt(m.Rt_ref_cam(), xc), x)
    for kw in (dict(rt_ref_cam=m.rt_ref_cam()), dict(Rt_cam_ref=m.Rt_cam_ref()), dict(Rt_ref_cam=m.Rt_ref_cam()),
               dict(extrinsics_rt_fromref=rt), dict(extrinsics_Rt_toref=m.Rt_ref_cam())):
        m2 = cameramodel(intrinsics=("LENSMODEL_OPENCV8", intr), imagersize=(4000, 2200), **kw)
        assert np.allclose(m2.rt_cam_ref(), rt, atol=1e-12)
    assert np.array_equal(cameramodel(intrinsics=("LENSMODEL_OPENCV8", intr), imagersize=(4000, 2200)).rt_cam_ref(), np.zeros(6))
    # text -> object -> text
    path = str(tmp_path / "a.cameramodel")
    m.valid_intrinsics_region(np.array(((5, 5), (100, 5), (100, 80), (5, 80.))))        # closed for us
    m.write(path, note="two\nlines")
    text = open(path).read()
    assert text.startswith("# two\n# lines\n{") and "'extrinsics'" in text and "'rt_cam_ref'" in text
    for src in (path, open(path), io.StringIO(text), m):
        m3 = cameramodel(src)
        assert np.allclose(m3.intrinsics()[1], intr, rtol=1e-9) and np.allclose(m3.rt_cam_ref(), rt)
        assert m3.valid_intrinsics_region().shape == (5, 2)
    assert str(cameramodel(path)) == str(m)
    # the reference's hand-written fixture style (test/data/cam0.opencv8.cameramodel): old key name, no trailing comma
    old = "{\n 'lensmodel': 'LENSMODEL_OPENCV8',\n # c\n 'intrinsics': [" + ",".join(f"{v}" for v in intr) + ",],\n" \
          " 'extrinsics': [ 2e-2, -3e-1, -1e-2,  1., 2, -3., ],\n\n 'imagersize': [ 4000, 2200 ]\n}\n"
    m4 = cameramodel(io.StringIO(old))
    assert np.array_equal(m4.rt_cam_ref(), rt) and np.array_equal(m4.intrinsics()[1], intr)
    # ancient key names
    m5 = cameramodel(io.StringIO(old.replace("lensmodel", "distortion_model").replace("LENSMODEL_", "DISTORTION_")))
    assert m5.intrinsics()[0] == "LENSMODEL_OPENCV8"
    # errors
    with pytest.raises(CameramodelParseException): cameramodel(io.StringIO("not a model"))
    with pytest.raises(CameramodelParseException): cameramodel(io.StringIO(old.replace("'imagersize': [ 4000, 2200 ]", "")))
    with pytest.raises(CameramodelParseException):
        cameramodel(io.StringIO(old.replace("'imagersize'", "'rt_cam_ref': [0,0,0,0,0,1], 'imagersize'")))
    with pytest.raises(CameramodelParseException): cameramodel(io.StringIO(old.replace("{\n", "{ 'icam_intrinsics': 0,\n", 1)))
    with pytest.raises(Exception): cameramodel(intrinsics=("LENSMODEL_OPENCV8", intr[:-1]), imagersize=(4000, 2200))
    with pytest.raises(Exception): cameramodel(intrinsics=("LENSMODEL_OPENCV8", intr), imagersize=(4000, -1))
    with pytest.raises(Exception): cameramodel(intrinsics=("LENSMODEL_OPENCV8", intr))
    with pytest.raises(Exception): cameramodel(intrinsics=("LENSMODEL_OPENCV8", intr), imagersize=(4, 2), rt_cam_ref=rt, rt_ref_cam=rt)
    with pytest.raises(Exception): cameramodel(path, imagersize=(4, 2))
    with pytest.raises(Exception): cameramodel()
    with pytest.raises(Exception): m.imagersize((3, 4))
    with pytest.raises(Exception): m.valid_intrinsics_region(np.zeros((3, 2)))
    with pytest.raises(NotImplementedError): m.write(path, cahvor=True)
    # new intrinsics drop the region
    m.intrinsics(("LENSMODEL_PINHOLE", intr[:4]))
    assert m.valid_intrinsics_region() is None and np.array_equal(m.imagersize(), (4000, 2200))


def test_optimization_inputs_round_trip(amd, ref_api):
    from mrcal_amd.cameramodel import cameramodel, _serialize_optimization_inputs, _deserialize_optimization_inputs
    from mrcal_amd.synthetic import make_calibration_problem
    oi = make_calibration_problem(ref_api, Ncameras=2, Nframes=5, lensmodel="LENSMODEL_OPENCV4", seed=3)[0]
    oi["points"] = None
    oi["do_apply_regularization_unity_cam01"] = False           # a default: left out of the file
    s = _serialize_optimization_inputs(oi)
    assert isinstance(s, bytes) and s.isascii()
    # on disk: the pre-2.5 names only
    import base64
    z = np.load(io.BytesIO(base64.b85decode(s)))
    assert "frames_rt_toref" in z and "extrinsics_rt_fromref" in z and "rt_ref_frame" not in z and "rt_cam_ref" not in z
    assert "do_apply_regularization_unity_cam01" not in z
    d = _deserialize_optimization_inputs(s)
    for k, v in oi.items():
        if k == "do_apply_regularization_unity_cam01": continue
        if isinstance(v, np.ndarray): assert np.array_equal(d[k], v) and d[k].dtype == v.dtype, k
        else:                         assert d[k] == v and type(d[k]) == type(v), (k, d[k], v)
    assert d["points"] is None
    assert d["frames_rt_toref"].startswith("ERROR:") and d["extrinsics_rt_fromref"].startswith("ERROR:")
    # what was read can be written again (the markers under the old names are skipped), and both generations of names are taken
    assert _serialize_optimization_inputs(d) == s
    legacy = dict(oi); legacy["frames_rt_toref"] = legacy.pop("rt_ref_frame"); legacy["extrinsics_rt_fromref"] = legacy.pop("rt_cam_ref")
    assert np.array_equal(_deserialize_optimization_inputs(_serialize_optimization_inputs(legacy))["rt_ref_frame"], oi["rt_ref_frame"])
    both = dict(oi); both["frames_rt_toref"] = oi["rt_ref_frame"] + 1.
    with pytest.raises(Exception): _serialize_optimization_inputs(both)
    # a model made from a solve: camera 1 sits at extrinsics 0
    m = cameramodel(optimization_inputs=oi, icam_intrinsics=1)
    assert m.icam_intrinsics() == 1 and m.icam_extrinsics() == 0
    assert np.array_equal(m.rt_cam_ref(), oi["rt_cam_ref"][0]) and np.array_equal(m.intrinsics()[1], oi["intrinsics"][1])
    assert not m._extrinsics_moved_since_calibration()
    m.rt_cam_ref(oi["rt_cam_ref"][0] + 1e-3)
    assert m._extrinsics_moved_since_calibration()
    m0 = cameramodel(optimization_inputs=oi, icam_intrinsics=0)
    assert m0.icam_extrinsics() == -1 and np.array_equal(m0.rt_cam_ref(), np.zeros(6))
    m2 = cameramodel(io.StringIO(str(m)))
    assert m2.icam_intrinsics() == 1 and m2.icam_extrinsics() == 0
    assert np.array_equal(m2.optimization_inputs()["observations_board"], oi["observations_board"])
    assert m2.optimization_inputs()["verbose"] is False and m2._optimization_inputs_match(m)
    m2.optimization_inputs_reset()
    assert m2.optimization_inputs() is None
    with pytest.raises(Exception): cameramodel(optimization_inputs=oi)
    with pytest.raises(Exception): cameramodel(icam_intrinsics=0)
    # intrinsics() with new numbers and no solve forgets the solve
    m.intrinsics(m.intrinsics())
    assert m.optimization_inputs() is None and m.icam_intrinsics() is None


# --------------------------------------------------------------------- real calibrations
REAL = ("real_opencv8-0", "real_splined-0")


def real_inputs(name):
    from mrcal_amd.cameramodel import cameramodel
    m = cameramodel(os.path.join(GOLDEN_DIR, name + ".cameramodel"))
    return m, m.optimization_inputs()


@pytest.mark.parametrize("name", REAL)
def test_real_calibration_parses_and_the_reference_takes_it(amd, ref_api, name):
    m, oi = real_inputs(name)
    assert oi["observations_board"].shape == (186, 10, 10, 3) and oi["intrinsics"].shape[0] == 1
    assert np.array_equal(m.imagersize(), (6016, 4016)) and m.icam_intrinsics() == 0 and m.icam_extrinsics() == -1
    assert np.allclose(m.intrinsics()[1], oi["intrinsics"][0], rtol=1e-9)
    assert m.valid_intrinsics_region() is not None
    b, x, J, _ = ref_api.optimizer_callback(**oi, no_factorization=True)
    Npix = int((oi["observations_board"][..., 2] > 0).sum())
    rms = np.sqrt((x[:2*186*100]**2).sum()/Npix)
    assert 0.1 < rms < 2.0, rms                     # a converged calibration of real data: sub-pixel to pixel-level fit
    # the stored point is the optimum of the stored problem: the gradient vanishes there
    g = J.T @ x
    assert np.abs(g).max() < 1e-3*np.abs(J.T @ np.abs(x)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", REAL)
def test_real_calibration_callback_matches_reference(amd, ref_api, name):
    _, oi = real_inputs(name)
    b, x, J, _ = amd.optimizer_callback(**oi, no_factorization=True)
    br, xr, Jr, _ = ref_api.optimizer_callback(**real_inputs(name)[1], no_factorization=True)
    assert np.array_equal(b, br)
    assert np.array_equal(J.indptr, Jr.indptr) and np.array_equal(J.indices, Jr.indices)
    assert relative_error(x, xr).max() < 1e-6
    assert relative_error(J.data, Jr.data).max() < 1e-6


@pytest.mark.gpu
def test_real_calibration_resolves_to_the_stored_optimum(amd, ref_api):
    from mrcal_amd.synthetic import copy_inputs
    _, oi = real_inputs("real_opencv8-0")
    oi["do_apply_outlier_rejection"] = False         # the stored weights already carry the outliers of the original solve
    # from the stored optimum: nothing to do, here and in the reference
    a, ar = copy_inputs(oi), copy_inputs(oi)
    s  = amd.optimize(**a)
    sr = ref_api.optimize(**ar)
    rms0 = sr["rms_reproj_error__pixels"]
    assert 0.1 < rms0 < 2.0
    assert abs(s["rms_reproj_error__pixels"] - rms0) < 1e-6
    assert s["Noutliers_board"] == sr["Noutliers_board"] == int((oi["observations_board"][..., 2] <= 0).sum())
    assert np.abs(a["intrinsics"] - oi["intrinsics"]).max() < 1e-3*np.abs(oi["intrinsics"]).max()
    # from a perturbed state: back to it, on the GPU and in the reference alike
    def perturbed():
        p = copy_inputs(oi)
        p["intrinsics"][:, :4] *= 1. + 0.01*np.array((1., -1., 0.3, -0.3))
        p["intrinsics"][:, 4:] *= 0.7
        p["rt_ref_frame"] = p["rt_ref_frame"] + np.random.default_rng(1).normal(size=p["rt_ref_frame"].shape)*np.array((1e-3,)*3 + (5e-3,)*3)
        return p
    pg, pr = perturbed(), perturbed()
    sg = amd.optimize(**pg)
    sr = ref_api.optimize(**pr)
    assert abs(sg["rms_reproj_error__pixels"] - sr["rms_reproj_error__pixels"]) < 1e-6
    assert abs(sg["rms_reproj_error__pixels"] - rms0) < 1e-5
    assert np.abs(pg["intrinsics"][0, :4] - oi["intrinsics"][0, :4]).max() < 0.05       # pixels
    assert np.abs(pg["intrinsics"] - pr["intrinsics"]).max() < 1e-3


def test_c_reader_and_writer_ignore_the_numeric_locale(amd, tmp_path):
    """numbers in a .cameramodel are Python literals whatever LC_NUMERIC says (a comma as the decimal separator
    would split every number for literal_eval), and only decimal literals are numbers (strtod would also take hex
    floats and 'inf')"""
    import locale, ctypes as C
    L = amd._lib.lib
    L.mrcal_read_cameramodel_string.restype = C.c_void_p
    L.mrcal_read_cameramodel_string.argtypes = [C.c_char_p, C.c_int]
    L.mrcal_free_cameramodel.argtypes = [C.POINTER(C.c_void_p)]
    L.mrcal_write_cameramodel_file.restype = C.c_bool
    L.mrcal_write_cameramodel_file.argtypes = [C.c_char_p, C.c_void_p]
    text = b"{'lensmodel': 'LENSMODEL_PINHOLE', 'intrinsics': [1000.5, 1001.25, 320.5, 240.75], " \
           b"'rt_cam_ref': [0.5, 0.25, 0.125, 1.5, 2.5, 3.5], 'imagersize': [640, 480]}"
    have = None
    for name in ("de_DE.UTF-8", "fr_FR.UTF-8", "de_DE", "fr_FR"):
        try:
            locale.setlocale(locale.LC_NUMERIC, name); have = name; break
        except locale.Error:
            pass
    try:
        m = C.c_void_p(L.mrcal_read_cameramodel_string(text, len(text)))
        assert m.value
        out = str(tmp_path / "m.cameramodel")
        assert L.mrcal_write_cameramodel_file(out.encode(), m)
        L.mrcal_free_cameramodel(C.byref(m))
    finally:
        locale.setlocale(locale.LC_NUMERIC, "C")
    back = amd.cameramodel(out)
    assert np.array_equal(back.intrinsics()[1], (1000.5, 1001.25, 320.5, 240.75))
    assert np.array_equal(back.rt_cam_ref(), (0.5, 0.25, 0.125, 1.5, 2.5, 3.5))
    # (without such a locale installed the test still holds the C-locale path)
    for bad in (b"0x1p3", b"inf", b"-inf", b"nan", b"1_000"):
        t = text.replace(b"1000.5", bad)
        assert not L.mrcal_read_cameramodel_string(t, len(t)), bad
