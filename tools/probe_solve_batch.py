"""CHOLMOD_factorization.solve_xt_JtJ_bt on BASELINE configuration 1's shape: time against the number of
right-hand sides in the call (they are solved side by side on the device)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=10, Nframes=1000, object_width_n=10, object_height_n=10,
                                 lensmodel="LENSMODEL_OPENCV8", seed=4)
b, x, J, f = mrcal_amd.optimizer_callback(**copy_inputs(oi))
N = J.shape[1]
for nrhs in (1, 8, 512, 4800):
    bt = np.random.default_rng(0).normal(size=(nrhs, N))
    f.solve_xt_JtJ_bt(bt)
    t0 = time.perf_counter(); xt = f.solve_xt_JtJ_bt(bt); dt = time.perf_counter() - t0
    print(f"opencv8 solve_xt_JtJ_bt, {nrhs} right-hand sides: {dt*1e3:.2f} ms, {dt/nrhs*1e6:.1f} us each")
