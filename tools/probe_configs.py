#!/usr/bin/env python3
"""Trial-step time and full-solve time of every BASELINE.json configuration on one
GPU (dev tool; the table in DESIGN.md section 6). The multi-GPU configurations (3, 4)
are run unsharded: they fit one MI355X"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs, CONFIG2_LENSMODEL
from mrcal_amd.resident import Problem

def board(**kw):
    return make_calibration_problem(mrcal_amd._api, object_width_n=10, object_height_n=10, seed=2, **kw)[0]
def sfm():
    from mrcal_amd.synthetic import make_sfm_problem as sfm_problem
    return sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=6, noise=0.3)[0]
def sfm_boards():
    from mrcal_amd.synthetic import make_sfm_problem as sfm_problem
    return sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=6, noise=0.3, Nboard_frames=400)[0]
CONFIGS = (("0: 1 cam x 40 frames OPENCV4",            lambda: board(Ncameras=1,  Nframes=40,   lensmodel="LENSMODEL_OPENCV4")),
           ("1: 4 cams x 400 frames OPENCV8",          lambda: board(Ncameras=4,  Nframes=400,  lensmodel="LENSMODEL_OPENCV8")),
           ("metric: 8 cams x 1000 frames OPENCV8",    lambda: board(Ncameras=8,  Nframes=1000, lensmodel="LENSMODEL_OPENCV8")),
           ("2: 1 cam x 800 frames SPLINED 30x20",     lambda: board(Ncameras=1,  Nframes=800,
                                                                      lensmodel=CONFIG2_LENSMODEL,
                                                                      do_optimize_intrinsics_core=False)),     # the core is redundant with the surface (mrcal's own recipe locks it)
           ("3: 16 cams x 2000 frames OPENCV8",        lambda: board(Ncameras=16, Nframes=2000, lensmodel="LENSMODEL_OPENCV8")),
           ("4: SfM, 4 cams, 20k triangulated points", sfm),
           ("5: SfM + boards, 4 cams, 20k triangulated points, 400 board frames", sfm_boards))
print("(board problems, the splined one included: make_calibration_problem(seed=2); the SfM ones sfm_problem(seed=6). bench.py's solve of the metric's problem is seed 0)")
print()
print("| configuration | Nstate | Nmeas | Nnz(J) | trial step | full solve (iterations, outlier passes) |")
print("|---|---|---|---|---|---|")
only = [a for a in sys.argv[1:]]          # e.g. "3": just that configuration (for rocprofv3)
for name, make in CONFIGS:
    if only and name.split(":")[0] not in only: continue
    oi = make()
    with Problem(**copy_inputs(oi)) as p:
        _, tr = p.run_steps(3, None); p.synchronize()
        t0 = time.perf_counter(); n, tr = p.run_steps(20, tr); p.synchronize()
        step_us = 1e6*(time.perf_counter() - t0)/20
        Nstate, Nmeas, Nnz = p.Nstate, p.Nmeas, p.Nnz
    with Problem(**copy_inputs(oi)) as p:
        p.synchronize()
        t0 = time.perf_counter(); s = p.solve(); p.synchronize()
        dt = time.perf_counter() - t0
    print(f"| {name} | {Nstate} | {Nmeas} | {Nnz} | {step_us:.0f} µs | {dt:.3f} s ({s['Niterations']}, {s['Noutlier_passes']}) |", flush=True)
