"""SURVEY.md section 8(b), level 3: the solver's plug-in point. INTEGRATION.md
section 3 shows a dogleg_callback_t over the resident tier of libmrcal_amd.so;
tests/opshim/opshim.c IS that callback. Here it is compiled, handed to the
checker's dogleg_optimize2() (the restated libdogleg, oracle/dogleg_restated.c)
with mrcal's solver settings, and the result compared with

  - the reference's own callback driven by the same solver (oracle/_ref:
    mrcal_optimize(), which does exactly mrcal.c:6435-6439)
  - the product's own device-side solve, mrcal_amd.optimize()

and the Jt == NULL form of the callback with an evaluation that has J."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

from conftest import ROOT, relative_error
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs

pytestmark = pytest.mark.gpu

HERE = os.path.join(ROOT, "tests", "opshim")


@pytest.fixture(scope="module")
def opshim(amd):
    so  = os.path.join(HERE, "libopshim.so")
    src = os.path.join(HERE, "opshim.c")
    dog = os.path.join(ROOT, "oracle", "_build", "liboracle_dogleg.so")
    if not os.path.exists(dog):
        pytest.skip("oracle/_build/liboracle_dogleg.so is not built (make -C oracle)")
    deps = [src, os.path.join(ROOT, "include", "mrcal_amd.h"), os.path.join(ROOT, "oracle", "stubs", "dogleg.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-Wall",
                               "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle", "stubs"),
                               "-o", so, src])
    # the symbols the shim leaves undefined come from the two libraries: both global before it is opened
    C.CDLL(amd._libpath, mode=C.RTLD_GLOBAL)
    C.CDLL(dog, mode=C.RTLD_GLOBAL)
    L = C.CDLL(so)
    L.opshim_optimize.restype  = C.c_double
    L.opshim_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.opshim_residuals_only.restype  = None
    L.opshim_residuals_only.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.opshim_counts.restype, L.opshim_counts.argtypes = None, [C.POINTER(C.c_int)]*2
    return L


@pytest.mark.parametrize("lensmodel,Ncam,Nf", (("LENSMODEL_OPENCV8", 2, 24), ("LENSMODEL_OPENCV4", 3, 8)))
def test_dogleg_callback_plugin(amd, ref_api, opshim, lensmodel, Ncam, Nf):
    from mrcal_amd.resident import Problem
    oi, _ = make_calibration_problem(amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel=lensmodel,
                                     object_width_n=7, object_height_n=6, seed=9, make_outliers=False)
    # one dogleg_optimize2() call: mrcal.c:6435-6439 once. (No gross outliers in the data then: with
    # them and without their rejection the valley floor is flat enough for two runs that differ in the
    # 12th digit of J to stop 1e-4 of the rms apart)
    oi["do_apply_outlier_rejection"] = False
    with Problem(**copy_inputs(oi)) as p:
        b = p.b_packed()
        Nstate, Nmeas, Nnz = p.Nstate, p.Nmeas, p.Nnz
        # the Jt == NULL evaluation against one with the Jacobian
        x0 = np.zeros(Nmeas)
        opshim.opshim_residuals_only(p.handle, b.ctypes.data, x0.ctypes.data)
        p.set_b_packed(b); p.evaluate(with_jacobian=True)
        assert np.array_equal(x0, p.x())
        # libdogleg's loop over the GPU callback
        norm2 = opshim.opshim_optimize(p.handle, b.ctypes.data, Nstate, Nmeas, Nnz, 0)
        n, n0 = C.c_int(0), C.c_int(0)
        opshim.opshim_counts(C.byref(n), C.byref(n0))
    assert norm2 > 0 and n.value > 3
    # the same loop over the reference's own callback
    orr = copy_inputs(oi)
    sr = ref_api.optimize(**orr)
    assert abs(np.sqrt(norm2/Nmeas) - sr["rms_reproj_error__pixels"]) < 1e-9*sr["rms_reproj_error__pixels"]
    # (both stop when a step is shorter than 1e-7, packed units, somewhere on a flat valley floor)
    assert np.abs(b - sr["b_packed"]).max() < 2e-5
    # and the product's own solver
    oa = copy_inputs(oi)
    sa = amd.optimize(**oa)
    assert abs(sa["rms_reproj_error__pixels"] - sr["rms_reproj_error__pixels"]) < 1e-7*sr["rms_reproj_error__pixels"]
    assert np.abs(sa["b_packed"] - b).max() < 2e-5
