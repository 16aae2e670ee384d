import sys, os
sys.path.insert(0, "/root/repo")
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8")
oi["do_apply_outlier_rejection"] = False
p = Problem(**oi)
print(p.solve())
