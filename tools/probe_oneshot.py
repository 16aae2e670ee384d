#!/usr/bin/env python3
"""Wall clock of the ONE-SHOT entry point: mrcal_amd.optimize(**optimization_inputs) from host arrays to host arrays
(problem creation, H2D, structure build, solve, D2H, teardown), next to the resident solve (dev tool)
usage: probe_oneshot.py [ns | 4 | 5]      the metric's problem (default) or BASELINE configuration 4 / 5 as bench.py makes them"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, make_sfm_problem, copy_inputs
from mrcal_amd.resident import Problem
config = sys.argv[1] if len(sys.argv) > 1 else "ns"
if config == "ns": oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", seed=0)
else:              oi = make_sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=6, noise=0.3, Nboard_frames=(400 if config == "5" else 0))[0]
for i in range(3):
    a = copy_inputs(oi)
    t0 = time.perf_counter(); s = mrcal_amd.optimize(**a); dt = time.perf_counter() - t0
    print(f"optimize() call {i}: {dt:.3f} s  rms {s['rms_reproj_error__pixels']:.5f}  outliers {s['Noutliers_board']}")
a = copy_inputs(oi)
for i in range(2):
    a = copy_inputs(oi)
    t0 = time.perf_counter(); p = Problem(**a); p.synchronize(); t1 = time.perf_counter(); s = p.solve(); p.synchronize(); t2 = time.perf_counter(); p.close(); t3 = time.perf_counter()
    print(f"resident: create {t1-t0:.4f} s  solve {t2-t1:.4f} s  close {t3-t2:.4f} s")
t0 = time.perf_counter(); r = mrcal_amd.optimizer_callback(**copy_inputs(oi)); dt = time.perf_counter() - t0
print(f"optimizer_callback() with J to the host: {dt:.3f} s")
