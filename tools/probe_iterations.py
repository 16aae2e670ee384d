#!/usr/bin/env python3
"""Iteration counts of the GPU dog-leg against the CPU checker's (restated
libdogleg driving the reference's callback) on the same problem (dev tool)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api  import Api
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
Ncam = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Nf   = int(sys.argv[2]) if len(sys.argv) > 2 else 50
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel="LENSMODEL_OPENCV8")
for rej in (False, True):
    o = copy_inputs(oi); o["do_apply_outlier_rejection"] = rej
    p = Problem(**o)
    st = p.solve()
    print("GPU  outlier_rejection", rej, st)
    ref = Api(MrcalLib(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libmrcal_ref.so")))
    o = copy_inputs(oi); o["do_apply_outlier_rejection"] = rej
    s = ref.optimize(**o)
    n = [C.c_int(0) for _ in range(3)]
    ref.clib.dogleg_restated_last_counts(*[C.byref(v) for v in n])
    print("CPU  outlier_rejection", rej, "rms", s["rms_reproj_error__pixels"], "Nout", s["Noutliers_board"], "last run: steps, callbacks, factorizations", [v.value for v in n])
