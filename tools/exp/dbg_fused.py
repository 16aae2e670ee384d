import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from mrcal_amd.resident import Problem
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=3, Nframes=10, lensmodel="LENSMODEL_OPENCV8", seed=1)
with Problem(**copy_inputs(oi)) as p:
    s = p.solve()
    print({k:s[k] for k in ("Niterations","Nevaluations","Nfactorizations","norm2_x","rms_reproj_error__pixels")})
