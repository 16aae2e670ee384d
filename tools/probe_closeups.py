#!/usr/bin/env python3
"""Step and solve time of a splined calibration whose boards are CLOSE-UPS (boxes of ~15 x 15 control points: 2 x 2
sub-boxes in the assembly, solver_kernels.hpp SPL_MAXSUB), beside BASELINE configuration 2's far boards (dev tool):
python tools/probe_closeups.py [Nframes [board_distance [object_spacing]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd.resident import Problem
Nframes  = int(sys.argv[1])   if len(sys.argv) > 1 else 800
distance = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
spacing  = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=Nframes, object_width_n=10, object_height_n=10,
                                lensmodel="LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=30_Ny=20_fov_x_deg=120", seed=4,
                                do_optimize_intrinsics_core=False, board_distance=distance, object_spacing=spacing)
p = Problem(**oi)
_, tr = p.run_steps(2, None); p.synchronize()
t0=time.perf_counter(); n,tr = p.run_steps(10, tr); p.synchronize(); dt=time.perf_counter()-t0
print(f"{Nframes} frames at {distance} m, spacing {spacing}: ms/step", 1e3*dt/10)
t0=time.perf_counter(); s = p.solve(); p.synchronize(); print("solve s", time.perf_counter()-t0, {k: s[k] for k in ("Niterations","Noutlier_passes","rms_reproj_error__pixels","Noutliers_board")})
