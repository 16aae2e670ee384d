#!/usr/bin/env python3
"""Ablation timing of the board kernel (dev tool). debug_ablate bits: 1 no J
copy-out, 2 no MFMA, 4 no projection arithmetic, 16 exit after the startup"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem
Ncam  = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Nf    = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lens  = sys.argv[3] if len(sys.argv) > 3 else "LENSMODEL_OPENCV8"
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel=lens)
p = Problem(**oi)
f = p._lib.mrcal_amd_problem_debug_time_evaluate
f.restype = C.c_double; f.argtypes = [C.c_void_p, C.c_bool, C.c_int, C.c_int]
for gram in (False, True):
    for ab in (0, 1, 2, 3, 4, 7, 16):
        if not gram and (ab & 2): continue
        ms = f(p.handle, gram, ab, 20)
        print(f"gram={int(gram)} ablate={ab:2d} {ms*1e3:8.1f} us")
