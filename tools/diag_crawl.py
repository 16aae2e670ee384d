#!/usr/bin/env python3
"""Is JtJ really numerically indefinite where the device Cholesky gives up?
(dev tool; the findings are in DESIGN.md section 6)

Runs the dog-leg one trial step at a time until lambda is first raised (= a
pivot <= 0 in the Schur path), then takes the block normal equations of the
current point to the host and asks LAPACK:

  - does numpy.linalg.cholesky factor the dense JtJ?  smallest / largest eigenvalue
  - does the Schur complement S = A - B D^-1 Bt, formed on the host in float64
    and in long double, factor? its smallest eigenvalue
  - where does an unpivoted host Cholesky of S (same algorithm as the device's)
    hit its first non-positive pivot

    python tools/diag_crawl.py ns|c1|c2|c3 [max_trials]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs
from mrcal_amd.resident import Problem

which = sys.argv[1] if len(sys.argv) > 1 else "ns"
max_trials = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg = dict(ns=dict(Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8", seed=0),
           c1=dict(Ncameras=4, Nframes=400,  lensmodel="LENSMODEL_OPENCV8", seed=2),
           c2=dict(Ncameras=1, Nframes=800,  lensmodel="LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=30_Ny=20_fov_x_deg=120",
                   seed=4, do_optimize_intrinsics_core=False),
           c3=dict(Ncameras=16, Nframes=2000, lensmodel="LENSMODEL_OPENCV8", seed=2))[which]
oi, _ = make_calibration_problem(mrcal_amd._api, object_width_n=10, object_height_n=10, **cfg)


def dense_from_blocks(ne, Nstate):
    Nie, NE, Nwarp, Nfb, NEb, Nc = ne["Nie"], ne["NE"], ne["Nwarp"], ne["Nfb"], ne["NEb"], ne["Nc"]
    iS = np.concatenate((np.arange(Nie), np.arange(Nie+NE, Nie+NE+Nwarp)))
    iE = np.arange(Nie, Nie+NE)
    N = np.zeros((Nstate, Nstate))
    N[np.ix_(iS, iS)] = ne["A"]
    N[np.ix_(iE, iS)] = ne["Bt"]
    N[np.ix_(iS, iE)] = ne["Bt"].T
    for b in range(NEb):
        if b < Nfb: e0, de = 6*b, 6
        else:       e0, de = 6*Nfb + 3*(b-Nfb), 3
        N[Nie+e0:Nie+e0+de, Nie+e0:Nie+e0+de] = ne["D"][b,:de,:de]
    return N, iS, iE


def schur(ne, dtype):
    A  = ne["A"].astype(dtype); Bt = ne["Bt"].astype(dtype)
    S  = A.copy()
    Nfb, NEb = ne["Nfb"], ne["NEb"]
    for b in range(NEb):
        if b < Nfb: e0, de = 6*b, 6
        else:       e0, de = 6*Nfb + 3*(b-Nfb), 3
        D = ne["D"][b,:de,:de].astype(dtype)
        # the device's route: L L^T = D, W = L^-1 Bt_e, S -= W^T W
        L = np.zeros((de,de), dtype=dtype)
        for j in range(de):
            d = D[j,j] - (L[j,:j]**2).sum()
            L[j,j] = np.sqrt(d)
            for i in range(j+1, de):
                L[i,j] = (D[i,j] - (L[i,:j]*L[j,:j]).sum())/L[j,j]
        W = np.zeros((de, A.shape[0]), dtype=dtype)
        for i in range(de):
            W[i] = (Bt[e0+i] - L[i,:i] @ W[:i])/L[i,i]
        S -= W.T @ W
    return S


def first_bad_pivot(S):
    S = np.array(S, dtype=np.float64)
    n = S.shape[0]
    L = np.zeros_like(S)
    for j in range(n):
        d = S[j,j] - (L[j,:j]**2).sum()
        if not d > 0: return j, d
        L[j,j] = np.sqrt(d)
        L[j+1:,j] = (S[j+1:,j] - L[j+1:,:j] @ L[j,:j])/L[j,j]
    return -1, 0.0


with Problem(**copy_inputs(oi)) as p:
    tr = None
    lam_prev = 0.0
    t0 = time.time()
    for it in range(max_trials):
        _, tr = p.run_steps(1, tr)
        st = p.solver_stats()
        if st["lambda_"] > lam_prev:
            print(f"trial {it}: lambda raised {lam_prev:g} -> {st['lambda_']:g}; |x|^2 {st['norm2_x']:.12g} trust region {tr:g} "
                  f"accepted {st['Niterations']} factorizations {st['Nfactorizations']}")
            break
    else:
        print(f"no factorization failure in {max_trials} trials; |x|^2 {p.solver_stats()['norm2_x']:.12g}")
        sys.exit(0)
    ne = p.normal_equations()
    print("blocks: Nc", ne["Nc"], "NE", ne["NE"], "NEb", ne["NEb"])
    N, iS, iE = dense_from_blocks(ne, p.Nstate)
    print("dense JtJ", N.shape, "symmetric to", np.abs(N - N.T).max()/np.abs(N).max())
    t0 = time.time()
    try:
        np.linalg.cholesky(N); ok = True
    except np.linalg.LinAlgError as e:
        ok = False
    print(f"LAPACK Cholesky of the dense JtJ: {'succeeds' if ok else 'FAILS'} ({time.time()-t0:.1f} s)")
    t0 = time.time()
    w_lo = scipy.linalg.eigvalsh(N, subset_by_index=[0, 4])
    w_hi = scipy.linalg.eigvalsh(N, subset_by_index=[p.Nstate-1, p.Nstate-1])
    print(f"eigenvalues of JtJ: smallest {w_lo}, largest {w_hi[0]:g}, cond {w_hi[0]/w_lo[0]:.3g} ({time.time()-t0:.1f} s)")
    d = np.sqrt(np.diag(N))
    Nn = N/d[:,None]/d[None,:]
    wn = scipy.linalg.eigvalsh(Nn, subset_by_index=[0, 2])
    print(f"after Jacobi scaling: smallest eigenvalues {wn} (largest <= {p.Nstate})")
    for dtype in (np.float64, np.longdouble):
        S = schur(ne, dtype)
        S64 = np.array(S, dtype=np.float64)
        ws = scipy.linalg.eigvalsh(S64, subset_by_index=[0, 2]); wl = scipy.linalg.eigvalsh(S64, subset_by_index=[S64.shape[0]-1, S64.shape[0]-1])
        j, dpiv = first_bad_pivot(S64)
        print(f"host Schur complement in {np.dtype(dtype).name}: smallest eigenvalues {ws}, largest {wl[0]:g}; "
              f"unpivoted Cholesky: {'ok' if j < 0 else f'pivot {j} = {dpiv:g}'}")
    S64 = np.array(schur(ne, np.float64)); Sld = np.array(schur(ne, np.longdouble), dtype=np.float64)
    print("host S float64 vs long double: max abs diff", np.abs(S64 - Sld).max(), "relative to max|A|", np.abs(S64 - Sld).max()/np.abs(ne["A"]).max())
    dS = np.sqrt(np.abs(np.diag(Sld)))
    print("diag(S) range", np.diag(Sld).min(), np.diag(Sld).max(), " diag(A) range", np.diag(ne["A"]).min(), np.diag(ne["A"]).max())
    # which variables carry the smallest eigenvector of JtJ
    w, V = scipy.linalg.eigh(N, subset_by_index=[0, 0])
    v = V[:,0]
    top = np.argsort(-np.abs(v))[:12]
    print("smallest eigenvector of JtJ: largest components (state index: value)", [(int(i), float(f"{v[i]:.3g}")) for i in top])
    # the device's own verdict at this point, host-driven: how much lambda does it need
    try:
        p.gauss_newton_step()
        print("device gauss_newton_step(): lambda now", p.solver_stats()["lambda_"])
    except RuntimeError as e:
        print("device gauss_newton_step() failed:", e)
