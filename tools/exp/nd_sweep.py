#!/usr/bin/env python3
"""configuration-2-like problems of several sizes and seeds solved; one line each (run with and without MRCAL_AMD_NO_ND=1 and compare) (dev tool)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs, CONFIG2_LENSMODEL
from mrcal_amd.resident import Problem
for Nf in (100, 400, 800):
    for seed in (1, 2, 3):
        oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=Nf, object_width_n=10, object_height_n=10,
                                         lensmodel=CONFIG2_LENSMODEL, seed=seed, do_optimize_intrinsics_core=False)
        with Problem(**copy_inputs(oi)) as p:
            s = p.solve()
            nd = p.dissection()
            b = p.b_packed()
            print("Nf %d seed %d: iterations %d outliers %d rms %.12f |b| %.9f  nd rounds %d active %d nA %d nB %d nS %d" %
                  (Nf, seed, s["Niterations"], s["Noutliers_board"], s["rms_reproj_error__pixels"], float(np.linalg.norm(b)),
                   nd["rounds"], nd["active"], nd["nA"], nd["nB"], nd["nS"]), flush=True)
