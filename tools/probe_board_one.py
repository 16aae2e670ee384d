#!/usr/bin/env python3
"""Runs the evaluation kernels a few times in one configuration (for rocprofv3 --pmc)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem
gram, ablate, nrep = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8")
p = Problem(**oi)
f = p._lib.mrcal_amd_problem_debug_time_evaluate
f.restype = C.c_double; f.argtypes = [C.c_void_p, C.c_bool, C.c_int, C.c_int]
print(f(p.handle, bool(gram), ablate, nrep)*1e3, "us")
