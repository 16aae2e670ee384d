#!/usr/bin/env python3
"""The splined calibration of the reference's documentation (tests/golden/real_splined-0.cameramodel: 1 camera, 186
frames of a 10 x 10 board, 6016 x 4016, LENSMODEL_SPLINED_STEREOGRAPHIC): the boxes of control points under its boards,
a trial step, and a solve from a perturbed state (dev tool)"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import mrcal_amd
from mrcal_amd.cameramodel import cameramodel
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import copy_inputs
m  = cameramodel(os.path.join(R, "tests", "golden", "real_splined-0.cameramodel"))
oi = m.optimization_inputs()
lm = oi["lensmodel"]
print(lm, oi["observations_board"].shape, "Nstate", mrcal_amd.num_states(**oi))
with Problem(**copy_inputs(oi)) as p:
    p.normal_equations()          # (an evaluation: the control points a row touches are written by it)
    J = p.J().tocsr()
    Nx = int(lm.split("Nx=")[1].split("_")[0])
    nk = (oi["intrinsics"].shape[1] - 4)//2
    ncore = mrcal_amd.num_states_intrinsics(**oi) - 2*nk
    print("J", J.shape, "core variables", ncore, "first row's columns", J[0].indices[:8])
    w = []
    for o in range(oi["observations_board"].shape[0]):
        rows = J[200*o:200*(o+1)]
        idx  = rows.indices[(rows.indices >= ncore) & (rows.indices < ncore + 2*nk)]
        if len(idx) == 0: continue
        k = (np.unique(idx) - ncore)//2
        w.append(((k % Nx).max() - (k % Nx).min() + 1, (k // Nx).max() - (k // Nx).min() + 1))
    w = np.array(w)
    print("boxes of control points: median %d x %d, largest %d x %d; over 109 control points: %d of %d observations" %
          (np.median(w[:,0]), np.median(w[:,1]), w[:,0].max(), w[:,1].max(), int((w[:,0]*w[:,1] > 109).sum()), len(w)))
    _, tr = p.run_steps(2, None); p.synchronize()
    for _ in range(3): _, tr = p.run_steps(10, tr)     # (the box loop above left the GPU idle for a second: its clocks are down)
    p.synchronize()
    t0 = time.perf_counter(); n, tr = p.run_steps(10, tr); p.synchronize(); dt = time.perf_counter() - t0
    print("trial step: %.3f ms" % (1e3*dt/10))
q = copy_inputs(oi)
q["do_apply_outlier_rejection"] = False
q["intrinsics"][:, 4:] *= 0.9
for i in range(2):
    a = copy_inputs(q)
    t0 = time.perf_counter(); s = mrcal_amd.optimize(**a); dt = time.perf_counter() - t0
    print("solve from a perturbed state: %.3f s, rms %.4f px" % (dt, s["rms_reproj_error__pixels"]))
