#!/usr/bin/env python3
"""three trial steps at the metric's size, for builds with -DASM_TS / -DCHOL_TS (kernel-internal cycle stamps on stdout; dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd.resident import Problem
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", seed=0)
with Problem(**oi) as p:
    p.run_steps(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    p.synchronize()
