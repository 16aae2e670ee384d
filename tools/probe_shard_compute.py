#!/usr/bin/env python3
"""What ONE rank of an N-rank solve computes per trial step, measured alone on one GPU (dev tool; the "sharded"
rows of DESIGN.md section 7's scaling table). Rank 0's shard of the metric's problem (frames [0, 1000/N)) runs the
sharded step through the protocol driver with a stand-in for the collectives: every sum over the ranks is replaced
by N times this rank's summand (the shards are statistically alike, so the normal equations, the steps and the
accept/reject pattern are those of the real solve to a few percent). What is NOT in these numbers: the two
all-reduces of a step (their stand-in is one small torch kernel each, listed separately by rocprofv3).
usage: probe_shard_compute.py N [cameras frames]      (wrap in rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mrcal_amd
from mrcal_amd import _api
from mrcal_amd.synthetic import make_calibration_problem
from mrcal_amd.resident import Problem
from mrcal_amd.parallel import GpuShard, ShardedDogleg, partition_frames

N     = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Ncam  = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Nfr   = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=Ncam, Nframes=Nfr, lensmodel="LENSMODEL_OPENCV8",
                                 object_width_n=10, object_height_n=10, seed=0)
p = _api._ingest(dict(oi), callback=False)
fr = partition_frames(p.c_board["iframe"].reshape(-1,1), p.Nframes, N)[0]
Nstate, Nmeas = _api._sizes(p)

class TimesN:
    """the stand-in: sum over N alike ranks = N x mine"""
    Ncollectives = 0
    def sum(self, t):
        if N > 1: t.mul_(float(N))
        self.Ncollectives += 1

problem = Problem(_shard=fr if N > 1 else None, _leader=True, **oi) if N > 1 else None
if N == 1:
    # (the single-GPU solve as it is: no stand-in, no protocol driver)
    with Problem(**oi) as q:
        q.run_steps(30, None); q.synchronize()
        t0 = time.perf_counter(); q.run_steps(50, None); q.synchronize()
        print(json.dumps(dict(N=1, frames=Nfr, step_us=round(1e6*(time.perf_counter() - t0)/50, 1))))
    sys.exit(0)
shard = GpuShard(problem, Nmeas, p.Nobservations_board*100, False)
d = ShardedDogleg(shard, TimesN())
n, tr = d.run(max_steps=30, check_termination=False)
problem.synchronize()
t0 = time.perf_counter()
n, tr = d.run(max_steps=50, check_termination=False, trustregion=tr)
problem.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(N=N, frames=fr[1] - fr[0], step_us=round(1e6*dt/50, 1), Nevaluations=d.stats["Nevaluations"],
                      note="one rank's compute + two small stand-in kernels per step; no collectives")))
problem.close()
