#!/usr/bin/env python3
"""the dissection's plan along a solve of configuration 2 (dev tool): Nframes [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs, CONFIG2_LENSMODEL
from mrcal_amd.resident import Problem
Nf = int(sys.argv[1]) if len(sys.argv) > 1 else 200
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=Nf, object_width_n=10, object_height_n=10,
                                 lensmodel=CONFIG2_LENSMODEL, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 4, do_optimize_intrinsics_core=False)
with Problem(**copy_inputs(oi)) as p:
    tr = None
    for k in range(12):
        n, tr = p.run_steps(1, tr); p.synchronize()
        print(k, p.dissection())
    s = p.solve()
    print("after solve", p.dissection(), s["Niterations"], s["rms_reproj_error__pixels"])
