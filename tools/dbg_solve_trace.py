import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from mrcal_amd.resident import Problem
which = sys.argv[1]
cfg = dict(ns=dict(Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", seed=0),
           c2=dict(Ncameras=1, Nframes=800,  lensmodel="LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=30_Ny=20_fov_x_deg=120", seed=4, do_optimize_intrinsics_core=False),
           c3=dict(Ncameras=16, Nframes=2000, lensmodel="LENSMODEL_OPENCV8", seed=2))[which]
oi,_ = make_calibration_problem(mrcal_amd._api, object_width_n=10, object_height_n=10, **cfg)
with Problem(**oi) as p:
    print(p.solve())
