#!/usr/bin/env python3
"""which blocks of the normal equations differ from J^T J for a moving-camera problem (dev tool)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import copy_inputs
from test_moving_camera import moving_camera_problem
from test_solver_parity import blocks_to_dense, dense_normal
ref_frame0 = len(sys.argv) > 1 and sys.argv[1] == "1"
oi = moving_camera_problem(mrcal_amd._api, 9, ref_frame0)
os.environ["MRCAL_AMD_ELIMINATE"] = "extrinsics"
with Problem(**copy_inputs(oi)) as p:
    ne = p.normal_equations()
    J, x = p.J(), p.x()
    print({k: ne[k] for k in ("Nc","NE","NEb","Nfb","S_split","S_shift","E_state0","eliminates")})
N, g = dense_normal(J, x)
Ng = blocks_to_dense(ne, p.Nstate)
Ni = 8; Nx = ne["NE"] if ne["eliminates"] == "extrinsics" else 6*len(oi["rt_cam_ref"])
names = [("intr",0,Ni), ("ext",Ni,Ni+Nx), ("rest",Ni+Nx,p.Nstate)]
for a,a0,a1 in names:
    for b,b0,b1 in names:
        d = np.abs(Ng[a0:a1,b0:b1] - N[a0:a1,b0:b1])
        if d.size: print(a, b, "max diff %.3g of %.3g" % (d.max(), np.abs(N[a0:a1,b0:b1]).max()), "at", np.unravel_index(d.argmax(), d.shape))
print("g diff", np.abs(ne["g"] - g).max(), np.abs(g).max())
np.set_printoptions(linewidth=200, precision=3)
for b in range(ne["Nfb"]):
    s = slice(Ni+6*b, Ni+6*b+6)
    print("block", b, "D diff", np.abs(Ng[s,s]-N[s,s]).max(), "gpu max", np.abs(Ng[s,s]).max(), "true max", np.abs(N[s,s]).max(),
          "| Bt diff", np.abs(Ng[s][:, :Ni]-N[s][:, :Ni]).max(), "| g diff", np.abs(ne["g"][s]-g[s]).max())
b = 2; s = slice(Ni+6*b, Ni+6*b+6)
print(Ng[s,s]); print(N[s,s])
