"""mrcal_optimize(check_gradient=true) and (verbose=true): mrcal.c:6291,
6600-6605. check_gradient: no solve; for every state variable libdogleg's
dogleg_testGradient() table - the reported gradient (a column of J) beside a
central difference of x - as vnlog on stdout. The product's table (device J and
x) against the one the reference's code + the restated libdogleg print for the
same inputs. verbose: the per-iteration trace and the regularization report on
stderr; the solve itself is unchanged."""
import os
import subprocess
import sys
import numpy as np
import pytest

from conftest import ROOT, REFLIB_PATH

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
which, mode = sys.argv[1], sys.argv[2]
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api  import Api
from mrcal_amd.synthetic import make_calibration_problem
api = mrcal_amd._api if which == "amd" else Api(MrcalLib(%(reflib)r))
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=2, Nframes=3, lensmodel="LENSMODEL_OPENCV4",
                                 object_width_n=4, object_height_n=3, seed=2, make_outliers=False)
if mode == "check_gradient":
    s = api.optimize(_check_gradient=True, **oi)
    import ctypes
    ctypes.CDLL(None).fflush(None)          # the table went through C stdio
    print("# rms", s["rms_reproj_error__pixels"], flush=True)
else:
    oi["verbose"] = True
    s = api.optimize(**oi)
    print("rms %%.12g" %% s["rms_reproj_error__pixels"])
'''


def _run(which, mode):
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, reflib=REFLIB_PATH), which, mode],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout, r.stderr


def _table(out):
    rows = [l.split() for l in out.splitlines() if l and not l.startswith("#")]
    odd  = [r for r in rows if len(r) != 6]
    assert not odd, odd[:5]
    return np.array(rows, dtype=float)


def test_check_gradient_table_matches_reference():
    if not os.path.exists(REFLIB_PATH):
        pytest.skip("oracle/_ref/libmrcal_ref.so is not built")
    out_a, _ = _run("amd", "check_gradient")
    out_r, _ = _run("ref", "check_gradient")
    assert out_a.splitlines()[0] == out_r.splitlines()[0] == \
        "# ivar imeasurement gradient_reported gradient_observed error error_relative"
    assert "# rms nan" in out_a and "# rms nan" in out_r          # sqrt(-1/Nmeasurements), like the reference
    ta, tr = _table(out_a), _table(out_r)
    assert ta.shape == tr.shape and ta.shape[1] == 6
    Nstate = int(ta[:,0].max()) + 1
    Nmeas  = int(ta[:,1].max()) + 1
    assert ta.shape[0] == Nstate*Nmeas
    assert np.array_equal(ta[:,:2], tr[:,:2])
    # gradient_reported: J; printed with 6 significant digits
    scale = np.abs(tr[:,2]).max()
    assert np.abs(ta[:,2] - tr[:,2]).max() < 2e-6*scale
    # gradient_observed: a central difference of step 1e-6 in both: the same up to the rounding of x
    assert np.abs(ta[:,3] - tr[:,3]).max() < 1e-4*scale
    # and the analytic gradient is right: reported ~ observed where it matters
    big = np.abs(tr[:,2]) > 1e-3*scale
    assert np.median(ta[big,5]) < 1e-5


def test_verbose_reports_and_does_not_change_the_solve(amd):
    from mrcal_amd.synthetic import make_calibration_problem
    out, err = _run("amd", "verbose")
    assert "trial" in err and "reg err ratio (distortion,centerpixel)" in err
    oi, _ = make_calibration_problem(amd._api, Ncameras=2, Nframes=3, lensmodel="LENSMODEL_OPENCV4",
                                     object_width_n=4, object_height_n=3, seed=2, make_outliers=False)
    s = amd.optimize(**oi)
    assert ("rms %.12g" % s["rms_reproj_error__pixels"]) in out


TRAJECTORY_SCRIPT = r'''
import sys, os, numpy as np
sys.path.insert(0, %(root)r)
which = sys.argv[1]
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api  import Api
from mrcal_amd.synthetic import make_calibration_problem
api = mrcal_amd._api if which == "amd" else Api(MrcalLib(%(reflib)r))
# (OPENCV8 with its 1 %% of gross outliers left in and outlier rejection off: a few dozen steps that mean something)
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=2, Nframes=30, lensmodel="LENSMODEL_OPENCV8",
                                 object_width_n=8, object_height_n=7, seed=11)
oi["do_apply_outlier_rejection"] = False
if which == "amd": oi["verbose"] = True
s = api.optimize(**oi)
print("rms %%.15g" %% s["rms_reproj_error__pixels"])
'''


def test_the_trajectory_is_the_restated_libdoglegs_step_for_step():
    """Not only the same optimum: the same SEQUENCE of trial points. The device-side dog-leg (eager Gauss-Newton, all
    decisions on the GPU: csrc/solver_device.hpp "dog-leg control") and the restated libdogleg (lazy Gauss-Newton, the
    published loop: oracle/dogleg_restated.c) driving the reference's own callback, on a well-conditioned calibration
    from the same seed: the cost after every trial step - accepted or rejected - and the trust region it was taken
    with, side by side from the two traces. (What this can pin is the restatement, not libdogleg itself: DESIGN.md 3)"""
    import re
    if not os.path.exists(REFLIB_PATH):
        pytest.skip("oracle/_ref/libmrcal_ref.so is not built")
    def run(which, env):
        r = subprocess.run([sys.executable, "-c", TRAJECTORY_SCRIPT % dict(root=ROOT, reflib=REFLIB_PATH), which],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout, r.stderr
    out_a, err_a = run("amd", {})
    out_r, err_r = run("ref", {"DOGLEG_RESTATED_TRACE": "1"})
    # the checker: "step N: norm2_x A -> B; expected improvement ..., got ...; rho R; trustregion T" per TRIAL
    ref = [(float(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)))
           for m in re.finditer(r"step \d+: norm2_x (\S+) -> (\S+); expected improvement \S+, got \S+; rho (\S+); trustregion (\S+)", err_r)]
    # the product: "trial N: accepted K tr T |x|^2 C ..." per trial, C = the cost of the CURRENT point after the trial
    # (the first line is the evaluation of the seed in the checker's count of steps? no: both count trial steps)
    ours = [(int(m.group(1)), float(m.group(2)), float(m.group(3)))
            for m in re.finditer(r"trial\s+\d+: accepted\s+(\d+) tr (\S+)\s+\|x\|\^2 (\S+)", err_a)]
    assert len(ref) > 8
    # the cost of the current point after each of the checker's trials: the new one if rho > 0, else the old one
    cost_ref = [b if rho > 0 else a for a, b, rho, tr in ref]
    naccepted_ref = np.cumsum([rho > 0 for a, b, rho, tr in ref])
    # ... for as long as the steps mean something: once a step changes the cost in its 10th digit and beyond, the sign
    # of the gain ratio - accept or reject - is rounding noise in either solver (they go on for a dozen such trials)
    n = 0
    while n < len(ref) and abs(ref[n][1] - ref[n][0]) > 1e-10*ref[n][0]: n += 1
    assert n >= 8, n
    # (how many trials either solver goes on for behind those is rounding noise too: whoever's termination test fires first)
    assert len(ours) >= n, (len(ours), n)
    naccepted_ref = naccepted_ref[:n]; cost_ref = cost_ref[:n]; ref = ref[:n]
    cost_a = np.array([c for k, tr, c in ours[:n]])
    assert np.array_equal(np.array([k for k, tr, c in ours[:n]]), naccepted_ref), "accept/reject decisions differ"
    # The costs along the way: equal to what two exact factorizations of the same JtJ (condition 1e12: OPENCV8) leave of
    # a step - observed 1.2e-7 relative after the first step (from the identical seed cost, 311135.2423000849 in
    # both), 4e-8 after the second, 1e-11 from the tenth on. (10 significant digits are printed on the product's side)
    assert np.abs(cost_a/np.array(cost_ref) - 1).max() < 1e-6, (cost_a, cost_ref)
    assert np.abs(cost_a[-3:]/np.array(cost_ref[-3:]) - 1).max() < 1e-9
    print(f"{n} trial steps compared, {int(naccepted_ref[-1])} of them accepted: cost {cost_ref[0]:.10g} -> {cost_ref[-1]:.10g}")
    # the trust region each trial left behind: the checker prints the one it was TAKEN with, ours the one after
    tr_after_ref = [tr for a, b, rho, tr in ref[1:]]
    tr_after_a   = [tr for k, tr, c in ours[:n-1]]
    assert np.abs(np.array(tr_after_a)/np.array(tr_after_ref) - 1).max() < 1e-3
    # and where they end: the same rms to 10 digits
    assert abs(float(out_a.split()[-1]) - float(out_r.split()[-1])) < 1e-10*float(out_r.split()[-1])
