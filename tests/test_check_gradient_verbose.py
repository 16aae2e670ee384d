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
