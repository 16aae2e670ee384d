#!/usr/bin/env python3
"""Runs the evaluation kernels a few times in one configuration (for rocprofv3 --pmc).
usage: probe_board_one.py gram ablate nrep [config]      config: ns (default), 1, 2, 3, 5 = BASELINE.json's configs[] as bench.py makes them"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem, make_sfm_problem, CONFIG2_LENSMODEL
gram, ablate, nrep = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
config = sys.argv[4] if len(sys.argv) > 4 else "ns"
def boards(**kw):
    return make_calibration_problem(mrcal_amd._api, object_width_n=10, object_height_n=10, seed=0, **kw)[0]
oi = dict(ns  = lambda: boards(Ncameras=8,  Nframes=1000, lensmodel="LENSMODEL_OPENCV8"),
          c1  = lambda: boards(Ncameras=4,  Nframes=400,  lensmodel="LENSMODEL_OPENCV8"),
          c2  = lambda: boards(Ncameras=1,  Nframes=800,  lensmodel=CONFIG2_LENSMODEL, do_optimize_intrinsics_core=False),
          c3  = lambda: boards(Ncameras=16, Nframes=2000, lensmodel="LENSMODEL_OPENCV8"),
          c5  = lambda: make_sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=6, noise=0.3, Nboard_frames=400)[0],
          )[config if config == "ns" else "c" + config]()
p = Problem(**oi)
f = p._lib.mrcal_amd_problem_debug_time_evaluate
f.restype = C.c_double; f.argtypes = [C.c_void_p, C.c_bool, C.c_int, C.c_int]
print(f(p.handle, bool(gram), ablate, nrep)*1e3, "us")
