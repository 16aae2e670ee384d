#!/usr/bin/env python3
"""Normal equations and the state after 1 and 4 trial steps of three small problems, to an .npz: run once with
each of two builds of libmrcal_amd.so and compare the files to see whether a kernel change moved any bits
(dev tool; usage: python tools/dbg_dump_normal.py out.npz)"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from mrcal_amd.resident import Problem
_partition = Problem.partition
def _tolerant(self):
    try: return _partition(self)
    except AttributeError: return {}          # (a build from before the partition query)
Problem.partition = _tolerant
out = {}
for name, kw in (("c3", dict(Ncameras=3, Nframes=11, lensmodel="LENSMODEL_OPENCV8")), ("c1", dict(Ncameras=1, Nframes=30, lensmodel="LENSMODEL_OPENCV8")),
                 ("c4", dict(Ncameras=4, Nframes=200, lensmodel="LENSMODEL_OPENCV4"))):
    oi,_ = make_calibration_problem(mrcal_amd._api, object_width_n=10, object_height_n=10, seed=5, **kw)
    with Problem(**copy_inputs(oi)) as p:
        ne = p.normal_equations()
        for k in ("A","Bt","D","g"): out[name+"_"+k] = np.array(ne[k])
        out[name+"_n2"] = ne["norm2_x"]
        p.run_steps(1)
        out[name+"_b1"] = p.b_packed().copy()
        p.run_steps(3)
        out[name+"_b4"] = p.b_packed().copy()
np.savez(sys.argv[1], **out)
