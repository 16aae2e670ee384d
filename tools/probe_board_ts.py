#!/usr/bin/env python3
"""Per-phase cycle profile of the board kernel (dev tool; needs a library built
with -DBOARD_TS: `bash mrcal_amd/csrc/build.sh -DBOARD_TS`)"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8")
p = Problem(**oi)
f = p._lib.mrcal_amd_problem_debug_timestamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_bool, C.c_void_p]
for gram in (True, False):
    out = np.zeros((8000, 8), dtype=np.int64)
    n = f(p.handle, gram, out.ctypes.data)
    t = out[:n]
    t0 = t[:,0].min()
    dur = t[:,6] - t[:,0]
    print(f"gram={gram}: kernel span {(t[:,6].max()-t0)} cycles; per wave: total {dur.mean():.0f} "
          f"startup {(t[:,1]-t[:,0]).mean():.0f} projection {t[:,2].mean():.0f} tilewrite {t[:,3].mean():.0f} "
          f"copyout {t[:,4].mean():.0f} gram {t[:,5].mean():.0f} tail {(t[:,6]-t[:,1]-t[:,2:6].sum(axis=1)).mean():.0f}")
    # start-time distribution: generations
    st = np.sort(t[:,0] - t0)
    print("   start times (cycles) at percentiles 0,25,50,75,100:", [int(st[int(q*(n-1))]) for q in (0,.25,.5,.75,1)])
    hw = t[:,7]
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
    print("   distinct (se,cu,simd):", len(set(zip(se.tolist(), cu.tolist(), simd.tolist()))))
