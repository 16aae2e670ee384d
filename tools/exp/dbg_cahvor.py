import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api import Api
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
import arbiter
ref = Api(MrcalLib("oracle/_ref/libmrcal_ref.so"))
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=2, Nframes=8, lensmodel="LENSMODEL_CAHVOR", object_width_n=10, object_height_n=10, seed=31)
oa, orr = copy_inputs(oi), copy_inputs(oi)
sa = mrcal_amd.optimize(**oa); sr = ref.optimize(**orr)
print("rms", sa["rms_reproj_error__pixels"], sr["rms_reproj_error__pixels"], "outliers", sa["Noutliers_board"], sr["Noutliers_board"])
print("db max", np.abs(sa["b_packed"] - sr["b_packed"]).max(), "dx max", np.abs(sa["x"] - sr["x"]).max())
for nm, o in (("product", oa), ("checker", orr)):
    st, cost, b = arbiter.stationarity(ref, o)
    print(nm, "stationarity", st, "cost", cost, "lsq gain", arbiter.least_squares_gain(ref, o))
