import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from mrcal_amd.resident import Problem
for (nc, nf, lm) in ((3, 10, "LENSMODEL_OPENCV8"), (8, 1000, "LENSMODEL_OPENCV8"), (2, 50, "LENSMODEL_OPENCV4")):
    oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=nc, Nframes=nf, lensmodel=lm, seed=1)
    with Problem(**copy_inputs(oi)) as p:
        for k in range(6):
            n, tr = p.run_steps(1, None if k == 0 else tr)
            st = p.solver_stats()
            x1, J1, b1 = p.x(), p.J(), p.b_packed()
            ne1 = p.normal_equations()      # (re-evaluates at the current state, host-driven: two launches)
            x2, J2 = p.x(), p.J()
            dx = np.abs(x1 - x2).max(); dJ = np.abs(J1.data - J2.data).max()
            nbad = int((J1.data != J2.data).sum())
            print(f"{nc}x{nf} {lm} step {k}: accepted {st['Niterations']} |dx| {dx:.3g} |dJ| {dJ:.3g} J entries that differ {nbad} of {J1.nnz}; x differ {int((x1 != x2).sum())}")
