import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd import CHOLMOD_factorization
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", seed=0)
_,_,J,_ = mrcal_amd.optimizer_callback(**oi, no_factorization=True)
part = (8*12 + 7*6, 1000, 0, 2)
for i in range(3):
    t0 = time.perf_counter(); F = CHOLMOD_factorization(J, _partition=part); dt = time.perf_counter() - t0
    print(("plain" if os.environ.get("MRCAL_AMD_PLAIN_ROW_SUMS") else "no rounding"), "CHOLMOD_factorization(J) at the metric's size: %.1f ms" % (1e3*dt), F.rcond())
