import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
oi, truth = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=9, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=10, object_height_n=10, seed=12, make_outliers=False)
oi.update(dict(do_optimize_intrinsics_core=False, do_optimize_intrinsics_distortions=False))
oi.update(do_optimize_calobject_warp=False, do_apply_regularization=False, do_apply_outlier_rejection=False, calobject_warp=None)
oi["verbose"]=False
try:
    s = mrcal_amd.optimize(**oi); print(s["rms_reproj_error__pixels"])
except Exception as e: print("EXC", e)
