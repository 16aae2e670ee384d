"""The metric's configuration (8 cameras x 1000 frames x 10x10, OPENCV8, everything optimized) SOLVED twice: by
mrcal_amd.optimize() on the GPU and by the reference's own mrcal_optimize() (oracle/_ref: mrcal.c compiled in
place, the restated libdogleg underneath) on one host core. ~3 minutes of CPU. Writes the comparison as JSON
(default profiles/r03_ns_solve_vs_reference.json); tests/test_full_size.py::test_solve_matches_reference_at_metric_size
asserts the same bounds in the suite.

    python tools/ns_solve_vs_reference.py [out.json]
"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api import Api
from mrcal_amd.synthetic import make_calibration_problem, copy_inputs

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_ns_solve_vs_reference.json")
ref = Api(MrcalLib(os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")))
oi, _ = make_calibration_problem(mrcal_amd._api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8",
                                 object_width_n=10, object_height_n=10, seed=2)
oa, orr = copy_inputs(oi), copy_inputs(oi)
t0 = time.time(); sa = mrcal_amd.optimize(**oa); ta = time.time() - t0
t0 = time.time(); sr = ref.optimize(**orr);      tr = time.time() - t0
db = np.abs(sa["b_packed"] - sr["b_packed"])
rec = dict(workload = "8 cameras x 1000 frames x 10x10 corners, LENSMODEL_OPENCV8, all variables, warp + regularization, outlier rejection (seed 2)",
           Nstate = int(sa["b_packed"].size), Nmeasurements = int(sa["x"].size),
           gpu = dict(seconds_optimize_call = ta, rms_reproj_error__pixels = sa["rms_reproj_error__pixels"], Noutliers_board = int(sa["Noutliers_board"]),
                      cost = float(sa["x"] @ sa["x"])),
           reference_cpu = dict(seconds_optimize_call = tr, cores = 1, rms_reproj_error__pixels = sr["rms_reproj_error__pixels"],
                                Noutliers_board = int(sr["Noutliers_board"]), cost = float(sr["x"] @ sr["x"]),
                                note = "the reference's mrcal_optimize() (mrcal.c compiled in place) over the restated libdogleg/Cholesky"),
           outlier_masks_identical = bool(np.array_equal(oa["observations_board"][...,2] < 0, orr["observations_board"][...,2] < 0)),
           rms_relative_difference = abs(sa["rms_reproj_error__pixels"] - sr["rms_reproj_error__pixels"])/sr["rms_reproj_error__pixels"],
           b_packed_max_abs_difference = float(db.max()), b_packed_argmax = int(db.argmax()),
           b_packed_differences_above_2e_5 = int((db > 2e-5).sum()),
           x_max_abs_difference = float(np.abs(sa["x"] - sr["x"]).max()))
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec, indent=1))
