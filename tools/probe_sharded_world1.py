import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd.parallel import ShardedProblem
from mrcal_amd.resident import Problem
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", object_width_n=10, object_height_n=10, seed=0)
for cls in (Problem, ShardedProblem):
    p = cls(**oi)
    _, tr = p.run_steps(5, None)
    p.synchronize()
    t0=time.perf_counter(); n,tr = p.run_steps(50, tr); p.synchronize(); dt=time.perf_counter()-t0
    print(cls.__name__, "ms/step", 1e3*dt/50, p.solver_stats())
    p.close()
