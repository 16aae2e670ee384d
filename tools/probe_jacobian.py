#!/usr/bin/env python3
"""Times the Jacobian build at a given problem size (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mrcal_amd
from mrcal_amd.resident import Problem
from mrcal_amd.synthetic import make_calibration_problem

Ncam  = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Nf    = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lens  = sys.argv[3] if len(sys.argv) > 3 else "LENSMODEL_OPENCV8"
t0 = time.time()
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=Ncam, Nframes=Nf, lensmodel=lens)
print(f"generated in {time.time()-t0:.2f}s")
p = Problem(**oi)
print("Nstate", p.Nstate, "Nmeas", p.Nmeas, "Nnz", p.Nnz)
for with_J in (True, False):
    for i in range(3): p.evaluate(with_J)
    ts = []
    kms = []
    for i in range(20):
        t0 = time.perf_counter()
        p.evaluate(with_J)
        ts.append(time.perf_counter()-t0)
        kms.append(p.jacobian_kernel_ms())
    NPTS = oi["observations_board"].shape[1]*oi["observations_board"].shape[2]
    bytes_alg = p.Nmeas//2*24 + p.Nmeas*8 + (p.Nnz*8 if with_J else 0)
    print(f"with_J={with_J}: wall min {min(ts)*1e3:.3f} ms median {np.median(ts)*1e3:.3f} ms; "
          f"board kernel min {min(kms):.4f} median {np.median(kms):.4f} ms; "
          f"alg bytes {bytes_alg/1e6:.1f} MB -> {bytes_alg/1e9/(np.median(kms)*1e-3) if kms[0]>0 else 0:.0f} GB/s")
