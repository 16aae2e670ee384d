import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api  import Api
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
Nf = int(sys.argv[1]); sp = float(sys.argv[2])
oi, truth = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=Nf,
                                     lensmodel="LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120",
                                     object_width_n=10, object_height_n=10, seed=33, seed_perturbation=sp)
oi["do_optimize_intrinsics_core"] = False
oi["do_apply_outlier_rejection"] = False
p = Problem(**copy_inputs(oi))
print("GPU", p.solve())
ref = Api(MrcalLib(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libmrcal_ref.so")))
s = ref.optimize(**copy_inputs(oi))
n = [C.c_int(0) for _ in range(3)]
ref.clib.dogleg_restated_last_counts(*[C.byref(v) for v in n])
print("CPU rms", s["rms_reproj_error__pixels"], "steps, callbacks, factorizations", [v.value for v in n])
