#!/usr/bin/env python3
"""BASELINE configuration 2: what a nested-dissection order of the control-point grid could take off the
launch-per-panel Cholesky's chain (dev tool, GPU). The boxes of control points under the boards come from J's
columns; a separator is a strip of grid columns (or rows) that no box straddles: every board then touches the strip
and ONE side only, and the two sides' panels could be factored side by side."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, CONFIG2_LENSMODEL
from mrcal_amd.resident import Problem
oi,_ = make_calibration_problem(mrcal_amd._api, Ncameras=1, Nframes=800, object_width_n=10, object_height_n=10,
                                lensmodel=CONFIG2_LENSMODEL, seed=4, do_optimize_intrinsics_core=False)
lm = oi["lensmodel"]
Nx = int(lm.split("Nx=")[1].split("_")[0]); Ny = int(lm.split("Ny=")[1].split("_")[0])
with Problem(**oi) as p:
    p.normal_equations()
    J = p.J().tocsr()
nk = Nx*Ny
Nintr = mrcal_amd.num_states_intrinsics(**oi)
ncore = Nintr - 2*nk
print("Nintrinsics", Nintr, "core", ncore, "J", J.shape)
Nobs = oi["observations_board"].shape[0]
boxes = []
covered = np.zeros((Ny, Nx), bool)
for o in range(Nobs):
    rows = J[200*o:200*(o+1)]
    idx  = rows.indices[(rows.indices >= ncore) & (rows.indices < ncore + 2*nk)]
    k = (np.unique(idx) - ncore)//2
    x, y = k % Nx, k // Nx
    boxes.append((x.min(), x.max(), y.min(), y.max()))
    covered[y.min():y.max()+1, x.min():x.max()+1] = True      # (the Schur complement couples the whole box)
boxes = np.array(boxes)
w = boxes[:,1]-boxes[:,0]+1; h = boxes[:,3]-boxes[:,2]+1
print("boxes: width median %d max %d, height median %d max %d" % (np.median(w), w.max(), np.median(h), h.max()))
print("covered control points %d of %d" % (covered.sum(), nk))
for r in covered: print("".join("#" if c else "." for c in r))
Ncam_other = (J.shape[1] - 6*800) - 2*nk      # what else is in the camera block
n_coupled = 2*covered.sum() + Ncam_other
print("coupled variables", n_coupled, "panels", -(-n_coupled//64))
def report(axis):
    lo, hi = (boxes[:,0], boxes[:,1]) if axis == 0 else (boxes[:,2], boxes[:,3])
    N = Nx if axis == 0 else Ny
    cnt = covered.sum(axis=0) if axis == 0 else covered.sum(axis=1)
    best = []
    for c0 in range(1, N-1):
        for c1 in range(c0, N-1):
            if np.any((lo < c0) & (hi > c1)): continue
            A = 2*cnt[:c0].sum(); B = 2*cnt[c1+1:].sum(); S = 2*cnt[c0:c1+1].sum() + Ncam_other
            a, b, s = -(-A//64), -(-B//64), -(-S//64)
            if a > b: a, b = b, a
            best.append((max(a + 2, b) + s, b + s, c0, c1, A, B, S, a, b, s))
            break          # (the narrowest strip from this c0)
    best.sort()
    for r in best[:6]:
        print("axis %d strip %d..%d: A %d B %d sep %d variables = %d + %d + %d panels; chain %d launches (a <= b-2 design), %d (a = b allowed)" %
              (axis, r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[0], r[1]))
report(0); report(1)
