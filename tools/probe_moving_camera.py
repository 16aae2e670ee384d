#!/usr/bin/env python3
"""A camera moving in front of a stationary board (N poses, one frame; tests/test_moving_camera.py builds it as the
reference's _apply_moving_ref does): the trial step and the full solve with the extrinsics eliminated (what the
library picks for such a problem) and with the frames eliminated (the stationary-camera partition, forced).
usage: probe_moving_camera.py [Nposes=500]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import copy_inputs
from test_moving_camera import moving_camera_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
out = {"Nposes": N}
for ref_frame0 in (True, False):
    oi = moving_camera_problem(mrcal_amd._api, N, ref_frame0, lensmodel="LENSMODEL_OPENCV8")
    oi["do_apply_outlier_rejection"] = True
    for what in ("extrinsics", "frames"):
        os.environ["MRCAL_AMD_ELIMINATE"] = what      # (before the first problem; mrcal_amd_set_elimination() is the API)
        with Problem(**copy_inputs(oi)) as p:
            part = p.partition()
            ne = p.normal_equations()
            p.run_steps(3); p.synchronize()
            t0 = time.perf_counter(); p.run_steps(20); p.synchronize(); t1 = time.perf_counter()
        with Problem(**copy_inputs(oi)) as p:
            t2 = time.perf_counter(); st = p.solve(); t3 = time.perf_counter()
        out[f"ref_frame0={ref_frame0} eliminate={what}"] = dict(eliminates=part["eliminates"], Nstate=p.Nstate, camera_block=ne["Nc"],
            trial_step_us=round((t1-t0)/20*1e6, 1), solve_s=round(t3-t2, 4), iterations=st["Niterations"],
            rms=st["rms_reproj_error__pixels"], Noutliers=st["Noutliers_board"])
print(json.dumps(out, indent=1))
