#!/usr/bin/env python3
"""dev: the disputed splined fuzz case (sweep 23, case 22) under the fallback's three regimes: hooks from argv, e.g. lchol_sweep=1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mrcal_amd
for kv in sys.argv[1:]:
    mrcal_amd.set_test_hook(kv.split("=")[0], int(kv.split("=")[1]))
import fuzz_parity
from mrcal_amd.synthetic import copy_inputs
from mrcal_amd.resident import Problem
for seed, icase in ((23, 22), (11, 29)):
    rng = np.random.RandomState(seed)
    for ic, what, oi, *_ in fuzz_parity.board_cases(icase + 1, rng, mrcal_amd._api):
        if ic == icase: break
    with Problem(**copy_inputs(oi)) as p:
        s = p.solve()
        print(sys.argv[1:], seed, icase, what, "rms", s["rms_reproj_error__pixels"], "outliers", s["Noutliers_board"], "iterations", s["Niterations"],
              "passes", s["Noutlier_passes"], "sweep", p.uses_sweep(), "ratio %.2e" % p.lchol_diag_ratio(), flush=True)
