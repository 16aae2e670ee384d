#!/usr/bin/env python3
"""CPU only. Where the restated libdogleg (oracle/dogleg_restated.c) first declares JtJ "not positive definite" on the
disputed splined problems of the fuzz sweeps (profiles/r03_fuzz_parity.txt: sweep seed 23 case 22; seed 11 cases 29,
120): is the matrix it was looking at positive definite for LAPACK (numpy.linalg.cholesky = dpotrf) and by its
eigenvalues? The problems are rebuilt with tools/fuzz_parity.py's generator driven by the REFERENCE's library (the
perfect pixels then differ from the GPU-made ones in their last digits; the question asked here does not care).

    python tools/diag_splined_pd.py [seed:case ...]
"""
import os, sys, glob, struct, tempfile
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api import Api
from mrcal_amd.synthetic import copy_inputs
import fuzz_parity

ref = Api(MrcalLib(os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")))


def case_of(seed, icase_wanted):
    rng = np.random.RandomState(seed)
    for icase, what, oi, *_ in fuzz_parity.board_cases(icase_wanted + 1, rng, ref):
        if icase == icase_wanted: return what, oi


def read_dump(path):
    with open(path, "rb") as f:
        n, m, nnz = struct.unpack("iii", f.read(12))
        lam, = struct.unpack("d", f.read(8))
        p = np.frombuffer(f.read(4*(m+1)), dtype=np.int32)
        i = np.frombuffer(f.read(4*nnz), dtype=np.int32)
        x = np.frombuffer(f.read(8*nnz), dtype=np.float64)
    return sp.csr_matrix((x, i, p), shape=(m, n)), lam


wanted = [a for a in sys.argv[1:] if ":" in a] or ["23:22", "11:29", "11:120"]
for w in wanted:
    seed, icase = (int(v) for v in w.split(":"))
    what, oi = case_of(seed, icase)
    with tempfile.TemporaryDirectory() as d:
        os.environ["DOGLEG_RESTATED_DUMP_NOTPD"] = os.path.join(d, "notpd_")
        o = copy_inputs(oi)
        s = ref.optimize(**o)
        del os.environ["DOGLEG_RESTATED_DUMP_NOTPD"]
        dumps = sorted(glob.glob(os.path.join(d, "notpd_*.bin")), key=lambda p: int(p.rsplit("_", 1)[1][:-4]))
        print(f"sweep {seed} {what}: reference + restated libdogleg: rms {s['rms_reproj_error__pixels']:.9g}, "
              f"{s['Noutliers_board']} outliers; 'not positive definite' reported {len(dumps)} times")
        for path in dumps[:6]:
            J, lam = read_dump(path)
            A = (J.T @ J).toarray() + lam*np.eye(J.shape[1])
            ev = np.linalg.eigvalsh(A)
            try:
                np.linalg.cholesky(A); potrf = "succeeds"
            except np.linalg.LinAlgError:
                potrf = "FAILS"
            # and in the order the GPU product factors in: frames last -> first (the Schur complement of the frame blocks)
            print(f"   lambda {lam:8.2g}: JtJ + lambda I ({A.shape[0]} x {A.shape[0]}): eigenvalues {ev[0]:.3e} .. {ev[-1]:.3e} "
                  f"(condition {ev[-1]/max(ev[0], 1e-300):.2e}, {int((ev <= 0).sum())} non-positive); LAPACK dpotrf {potrf}")
