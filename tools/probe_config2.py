#!/usr/bin/env python3
"""Step time of BASELINE config 2 (splined 30x20, 800 frames, 1 camera) (dev tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, CONFIG2_LENSMODEL
from mrcal_amd.resident import Problem
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=800, object_width_n=10, object_height_n=10,
                                lensmodel=CONFIG2_LENSMODEL, seed=4,
                                do_optimize_intrinsics_core=False)
p = Problem(**oi)
_, tr = p.run_steps(2, None); p.synchronize()
t0=time.perf_counter(); n,tr = p.run_steps(10, tr); p.synchronize(); dt=time.perf_counter()-t0
print("config2 ms/step", 1e3*dt/10, p.solver_stats())
t0=time.perf_counter(); s = p.solve(); p.synchronize(); print("solve s", time.perf_counter()-t0, s)
