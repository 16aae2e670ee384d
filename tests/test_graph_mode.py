"""MRCAL_AMD_GRAPH=1: the trial step replayed as one captured hipGraph instead of
eight eager launches (solver.cpp queue_trial_step). Same kernels, same order, no
atomics anywhere: the solve must come out BIT-identical to the eager one. The
switch is read once per process, so each mode runs in a process of its own."""
import os
import subprocess
import sys
import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import mrcal_amd
from mrcal_amd.synthetic import make_calibration_problem
lensmodel = sys.argv[3]
extra = {"do_optimize_intrinsics_core": False} if "SPLINED" in lensmodel else {}
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=3, Nframes=20, lensmodel=lensmodel, seed=9, **extra)
s = mrcal_amd.optimize(**oi)
np.savez(sys.argv[2], b=s["b_packed"], x=s["x"], rms=s["rms_reproj_error__pixels"], Noutliers=s["Noutliers_board"],
         intrinsics=oi["intrinsics"], rt_ref_frame=oi["rt_ref_frame"])
"""


def run(tmp_path, graph, lensmodel):
    out = str(tmp_path / f"graph{int(graph)}.npz")
    env = dict(os.environ)
    env.pop("MRCAL_AMD_GRAPH", None)
    if graph: env["MRCAL_AMD_GRAPH"] = "1"
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, out, lensmodel], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


# (the splined models: their assembly forks to a second stream and joins again, captured into the graph too)
@pytest.mark.parametrize("lensmodel", ("LENSMODEL_OPENCV8", "LENSMODEL_SPLINED_STEREOGRAPHIC_order=3_Nx=11_Ny=8_fov_x_deg=120"))
def test_graph_replay_is_the_eager_solve(tmp_path, lensmodel):
    eager, graph = run(tmp_path, False, lensmodel), run(tmp_path, True, lensmodel)
    assert int(eager["Noutliers"]) == int(graph["Noutliers"])
    for k in ("b", "x", "rms", "intrinsics", "rt_ref_frame"):
        assert np.array_equal(eager[k], graph[k]), k
