import os, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api import Api
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from mrcal_amd.resident import Problem
ref = Api(MrcalLib(os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")))
lens = "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=16_Ny=12_fov_x_deg=120"
for Ncam in (2, 3):
    oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=Ncam, Nframes=60, lensmodel=lens, object_width_n=10, object_height_n=10, seed=11,
                                     do_optimize_intrinsics_core=False)
    print("Nstate", mrcal_amd.num_states(**oi), "Nintrinsics states", mrcal_amd.num_states_intrinsics(**oi))
    a, r = copy_inputs(oi), copy_inputs(oi)
    sa = mrcal_amd.optimize(**a); sr = ref.optimize(**r)
    print(Ncam, "cams: ours rms %.9f outliers %d | reference rms %.9f outliers %d" % (sa["rms_reproj_error__pixels"], sa["Noutliers_board"], sr["rms_reproj_error__pixels"], sr["Noutliers_board"]))
    with Problem(**copy_inputs(oi)) as p:
        p.solve(); print("   dissection", p.dissection())
