"""The two host-side solvers that bench.py's cpu_baseline times beside the reference's callback (oracle/schur_numpy.py:
the product's Schur-complement algorithm in numpy/LAPACK, and the same blocks through SciPy's SuperLU) against dense
numpy on the reference's own Jacobian. CPU only."""
import os
import sys
import numpy as np
import pytest

from conftest import ROOT
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs

sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.mark.parametrize("Ncameras,Nframes,lensmodel", ((1, 7, "LENSMODEL_OPENCV4"), (3, 9, "LENSMODEL_OPENCV8")))
def test_host_schur_and_superlu_match_the_dense_solve(ref_api, Ncameras, Nframes, lensmodel):
    import schur_numpy as sn
    oi, _ = make_calibration_problem(ref_api, Ncameras=Ncameras, Nframes=Nframes, lensmodel=lensmodel,
                                     object_width_n=7, object_height_n=6, seed=17)
    oi["observations_board"][1,2,3,2] = -1.
    b, x, J, _ = ref_api.optimizer_callback(no_factorization=True, **copy_inputs(oi))
    lay = sn.BoardLayout(ref_api, oi)
    A, Bt, D, gS, gE = sn.normal_equations(J, x, lay)
    Jd = J.toarray(); N = Jd.T @ Jd; g = Jd.T @ x
    S, f0 = lay.S_states, lay.i_frames
    assert np.abs(A - N[np.ix_(S, S)]).max() < 1e-10*np.abs(N).max()
    assert np.abs(gS - g[S]).max() < 1e-10*np.abs(g).max()
    for f in range(lay.Nframes):
        e = slice(f0 + 6*f, f0 + 6*f + 6)
        assert np.abs(D[f] - N[e, e]).max() < 1e-10*np.abs(N).max()
        assert np.abs(Bt[f] - N[e][:, S]).max() < 1e-10*np.abs(N).max()
    d0 = -np.linalg.solve(N, g)
    for solver in (sn.gauss_newton_step_schur, sn.gauss_newton_step_superlu):
        d = solver(J, x, lay)
        assert np.abs(N @ d + g).max() < 1e-7*np.abs(g).max()
        assert np.abs(d - d0).max() < 1e-5*max(1.0, np.abs(d0).max())
    # ... and the loop bench.py times makes progress from the seed
    r = sn.timed_trial_steps(ref_api, oi, 6, sn.gauss_newton_step_schur, copy_inputs)
    assert r["Ntrials"] == 6 and r["cost1"] < r["cost0"]
