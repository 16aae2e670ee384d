#!/usr/bin/env python3
"""assemble_factor_kernel without the elimination (the host-driven evaluation of the normal equations) under rocprofv3 (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd.resident import Problem
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", seed=0)
with Problem(**oi) as p:
    for i in range(12): p.normal_equations()
