#!/usr/bin/env python3
"""Times the solve at a given problem size (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem

Ncam  = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Nf    = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lens  = sys.argv[3] if len(sys.argv) > 3 else "LENSMODEL_OPENCV8"
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel=lens)
p = Problem(**oi)
print("Nstate", p.Nstate, "Nmeas", p.Nmeas, "Nnz", p.Nnz)
t0 = time.perf_counter()
st = p.solve()
t1 = time.perf_counter()
print("solve:", st, f"wall {t1-t0:.4f}s")
print(f"  -> {st['Nevaluations']/(t1-t0):.1f} evaluations/s")
b0 = p.b_packed()
# fixed number of steps from the seed
p2 = Problem(**oi)
p2.run_steps(3)
for K in (10, 20):
    t0 = time.perf_counter()
    n, tr = p2.run_steps(K)
    t1 = time.perf_counter()
    print(f"run_steps({K}): {(t1-t0)/K*1e3:.3f} ms/step, {K/(t1-t0):.1f} steps/s, trustregion {tr:g}, stats {p2.solver_stats()}")
