#!/usr/bin/env python3
"""How long the HOST takes to queue one sharded trial step (dev tool): if that
exceeds the GPU time of a step the multi-GPU run is host-bound"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29877", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl")
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd.parallel import ShardedProblem
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", object_width_n=10, object_height_n=10, seed=0)
sp = ShardedProblem(_always_communicate=True, **oi)
_, tr = sp.run_steps(5, None)
sp.synchronize()
d = sp.dogleg
with d.s.context():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50):
        d._queue(False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    sp.synchronize()
    t2 = time.perf_counter()
print(f"host queueing {1e6*(t1-t0)/50:.1f} us/step; until the GPU is done {1e6*(t2-t0)/50:.1f} us/step")
sp.close()
dist.destroy_process_group()
