#!/usr/bin/env python3
"""Per-phase cycle profile of the board kernel (dev tool; needs a library built
with -DBOARD_TS: `bash mrcal_amd/csrc/build.sh -DBOARD_TS`, run with MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so).
usage: probe_board_ts.py [config]     config: ns (default), 1, 3, 5 as tools/probe_board_one.py"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem, make_sfm_problem
config = sys.argv[1] if len(sys.argv) > 1 else "ns"
def boards(**kw):
    return make_calibration_problem(mrcal_amd._api, object_width_n=10, object_height_n=10, seed=0, **kw)[0]
oi = dict(ns  = lambda: boards(Ncameras=8,  Nframes=1000, lensmodel="LENSMODEL_OPENCV8"),
          c1  = lambda: boards(Ncameras=4,  Nframes=400,  lensmodel="LENSMODEL_OPENCV8"),
          c3  = lambda: boards(Ncameras=16, Nframes=2000, lensmodel="LENSMODEL_OPENCV8"),
          c5  = lambda: make_sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=6, noise=0.3, Nboard_frames=400)[0],
          )[config if config == "ns" else "c" + config]()
p = Problem(**oi)
Nobs = oi["observations_board"].shape[0]
f = p._lib.mrcal_amd_problem_debug_timestamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_bool, C.c_void_p]
ft = p._lib.mrcal_amd_problem_debug_time_evaluate
ft.restype = C.c_double; ft.argtypes = [C.c_void_p, C.c_bool, C.c_int, C.c_int]
for gram in (True, False):
    print(f"config {config}, {Nobs} observations, gram={gram}: evaluation launches event-timed {ft(p.handle, gram, 0, 5)*1e3:.1f} us")
    out = np.zeros((Nobs, 10), dtype=np.int64)
    n = f(p.handle, gram, out.ctypes.data)
    t = out[:n]
    t0 = t[:,0].min()
    dur = t[:,6] - t[:,0]
    print(f"   kernel span {(t[:,6].max()-t0)} cycles; per wave: total {dur.mean():.0f} (min {dur.min()}, max {dur.max()}) "
          f"startup {(t[:,1]-t[:,0]).mean():.0f} projection {t[:,2].mean():.0f} tilewrite {t[:,3].mean():.0f} "
          f"copyout {t[:,4].mean():.0f} gram {t[:,5].mean():.0f} tail {(t[:,6]-t[:,1]-t[:,2:6].sum(axis=1)).mean():.0f}")
    # the wall clock (100 MHz, one for the chip): when do the waves start and end?
    w0 = t[:,8].min()
    st = np.sort(t[:,8] - w0)*0.01; en = np.sort(t[:,9] - w0)*0.01
    print(f"   wall clock: first start -> last end {en[-1]:.2f} us")
    print("   start times (us) at percentiles 0,10,25,50,75,90,99,100:", [round(float(st[int(q*(n-1))]), 2) for q in (0,.1,.25,.5,.75,.9,.99,1)])
    print("   end   times (us) at percentiles 0,10,25,50,75,90,99,100:", [round(float(en[int(q*(n-1))]), 2) for q in (0,.1,.25,.5,.75,.9,.99,1)])
    life = (t[:,9] - t[:,8])*0.01
    print(f"   a wave's life (us): mean {life.mean():.2f} min {life.min():.2f} max {life.max():.2f}")
    hw = t[:,7] & 0xffffffff; xcc = (t[:,7] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; wv = hw & 0xf
    key = ((xcc*8 + se)*16 + cu)*4 + simd
    u, cnt = np.unique(key, return_counts=True)
    print(f"   distinct (xcc,se,cu,simd): {len(u)}; waves per SIMD: min {cnt.min()} max {cnt.max()}; histogram {np.bincount(cnt).tolist()}")
    cukey = (xcc*8 + se)*16 + cu
    uc, cc = np.unique(cukey, return_counts=True)
    print(f"   distinct CUs: {len(uc)}; waves per CU: min {cc.min()} max {cc.max()}; histogram {np.bincount(cc).tolist()}; per XCD {np.bincount(xcc).tolist()}")
    # concurrency: how many waves of the same SIMD overlap a wave's life, and does that lengthen it?
    order_k = np.argsort(key, kind='stable')
    conc = np.zeros(n)
    for k in u:
        idx = np.nonzero(key == k)[0]
        for i in idx:
            ov = np.minimum(t[idx,9], t[i,9]) - np.maximum(t[idx,8], t[i,8])
            conc[i] = np.clip(ov, 0, None).sum()/max(t[i,9] - t[i,8], 1)
    for lo, hi in ((0.9, 1.2), (1.2, 1.7), (1.7, 2.2), (2.2, 9)):
        m = (conc >= lo) & (conc < hi)
        if m.any(): print(f"   waves with {lo}-{hi} waves' worth of company on their SIMD (themselves included): {int(m.sum())}, life {life[m].mean():.2f} us")
    # duration against the start time: do the late starters run shorter (alone on their SIMD)?
    order = np.argsort(t[:,0])
    q = max(n//8, 1)
    print("   mean duration by start-time octile:", [int(dur[order[i*q:(i+1)*q]].mean()) for i in range(8)])
