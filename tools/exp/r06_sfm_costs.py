#!/usr/bin/env python3
"""Where the drop-in calls of a structure-from-motion problem (BASELINE configuration 4 / 5) spend their time (dev tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mrcal_amd
from mrcal_amd.synthetic import make_sfm_problem, copy_inputs
from mrcal_amd.resident import Problem
config = sys.argv[1] if len(sys.argv) > 1 else "4"
oi = make_sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=6, noise=0.3, Nboard_frames=(400 if config == "5" else 0))[0]
def t(f, n=3):
    best = 1e9
    for i in range(n):
        a = copy_inputs(oi); t0 = time.perf_counter(); r = f(a); best = min(best, time.perf_counter() - t0)
    return best
print("optimizer_callback(no_jacobian, no_factorization): %.4f s" % t(lambda a: mrcal_amd.optimizer_callback(**a, no_jacobian=True, no_factorization=True)))
print("optimizer_callback(no_factorization):              %.4f s" % t(lambda a: mrcal_amd.optimizer_callback(**a, no_factorization=True)))
print("optimizer_callback():                              %.4f s" % t(lambda a: mrcal_amd.optimizer_callback(**a)))
os.environ["MRCAL_AMD_DEBUG_CREATE"] = "1"
a = copy_inputs(oi); t0 = time.perf_counter(); p = Problem(**a); p.synchronize(); print("create %.4f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); p.evaluate(with_jacobian=True); p.synchronize(); print("first evaluate %.4f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); s = p.solve(); p.synchronize(); print("solve %.4f s" % (time.perf_counter() - t0))
p.close()
