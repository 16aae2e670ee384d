"""CHOLMOD_factorization.solve_xt_JtJ_bt on BASELINE configuration 2's shape (camera block 1206): time per
right-hand side.  rocprofv3 --kernel-trace --stats -- python tools/probe_splined_solve.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=800, object_width_n=10, object_height_n=10,
                                 lensmodel="LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=30_Ny=20_fov_x_deg=120",
                                 seed=4, do_optimize_intrinsics_core=False)
for _ in range(2):
    t0 = time.perf_counter(); b, x, J, f = mrcal_amd.optimizer_callback(**copy_inputs(oi))
    print("callback+factorization", time.perf_counter() - t0)
N = J.shape[1]
for nrhs in (1, 8, 512, 4800):
    bt = np.random.default_rng(0).normal(size=(nrhs, N))
    f.solve_xt_JtJ_bt(bt)
    t0 = time.perf_counter(); xt = f.solve_xt_JtJ_bt(bt); dt = time.perf_counter() - t0
    print(f"splined solve_xt_JtJ_bt, {nrhs} right-hand sides: {dt*1e3:.2f} ms, {dt/nrhs*1e6:.1f} us each")
JtJ = (J.T @ J)
res = (JtJ @ xt.T).T - bt
print("residual", np.abs(res).max(), "scale", np.abs(bt).max())
