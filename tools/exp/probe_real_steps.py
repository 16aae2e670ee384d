import os, sys, time
import numpy as np
R = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R)
import mrcal_amd
from mrcal_amd.cameramodel import cameramodel
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import copy_inputs
m  = cameramodel(os.path.join(R, "tests", "golden", "real_splined-0.cameramodel"))
oi = m.optimization_inputs()
for variant in ("steps only", "normal_equations first", "J first"):
    with Problem(**copy_inputs(oi)) as p:
        if variant == "normal_equations first": p.normal_equations()
        if variant == "J first": p.J()
        _, tr = p.run_steps(2, None); p.synchronize()
        for rep in range(2):
            t0 = time.perf_counter(); n, tr = p.run_steps(10, tr); p.synchronize(); dt = time.perf_counter() - t0
            print(variant, "trial step: %.3f ms" % (1e3*dt/10), n, p.solver_stats())
