#!/usr/bin/env python3
"""The reference's own mrcal_optimize() (oracle/_ref/libmrcal_ref.so: mrcal.c compiled in place, the restated libdogleg
underneath) run HERE on the two BASELINE.json configurations whose solve is too long for the GPU suite, its results
kept as small fixtures:

    config3   16 cameras x 2000 frames x 10x10, LENSMODEL_OPENCV8, everything optimized, outlier rejection (seed 2)
    config5   SfM: 4 cameras OPENCV4 (intrinsics locked) + 20000 triangulated points + 400 board frames (seed 9)
    ns_seed0  the metric's 8 cameras x 1000 frames OPENCV8 at bench.py's seed 0: what the 8-rank solve is compared on

    python tests/golden/make_reference_solves.py config5 config3       (CPU only; config3 takes ~an hour of one core)

The inputs are synthesized with the REFERENCE's library behind the Api (make_calibration_problem projects its perfect
corners through api.optimizer_callback()), so the GPU suite can make the same inputs on the GPU box from the same
library (tests/test_full_size.py::test_solve_matches_the_references_recorded_solve) - a hash of them is in the record.
A fixture is data: the outlier mask (packed bits), b_packed, rms, cost, counts - no reference source."""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api  import Api
from mrcal_amd.synthetic import make_calibration_problem, make_sfm_problem, copy_inputs

def recorded_inputs(name, api):
    if name == "config3":
        return make_calibration_problem(api, Ncameras=16, Nframes=2000, lensmodel="LENSMODEL_OPENCV8",
                                        object_width_n=10, object_height_n=10, seed=2)[0]
    if name == "ns_seed0":  # the metric's own problem, bench.py's seed: what the sharded solves are compared on
        return make_calibration_problem(api, Ncameras=8, Nframes=1000, lensmodel="LENSMODEL_OPENCV8",
                                        object_width_n=10, object_height_n=10, seed=0)[0]
    if name == "config1":   # (a small one, to try the machinery)
        return make_calibration_problem(api, Ncameras=4, Nframes=400, lensmodel="LENSMODEL_OPENCV8",
                                        object_width_n=10, object_height_n=10, seed=2)[0]
    if name == "config5":
        oi = make_sfm_problem("LENSMODEL_OPENCV4", Ncam=4, Npoints=20000, seed=9, noise=0.3, Nboard_frames=400)[0]
        oi["do_apply_regularization_unity_cam01"] = True
        oi["do_apply_outlier_rejection"] = True
        return oi
    raise ValueError(name)

def inputs_hash(oi):
    h = hashlib.sha256()
    for k in sorted(oi):
        v = oi[k]
        if isinstance(v, np.ndarray):
            h.update(k.encode()); h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()

if __name__ == "__main__":
    ref = Api(MrcalLib(os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")))
    for name in sys.argv[1:]:
        oi = recorded_inputs(name, ref)
        h  = inputs_hash(oi)
        o  = copy_inputs(oi)
        t0 = time.time(); s = ref.optimize(**o); dt = time.time() - t0
        rec = dict(inputs_sha256 = h, seconds = dt,
                   b_packed = s["b_packed"], rms_reproj_error__pixels = s["rms_reproj_error__pixels"],
                   cost = float(s["x"] @ s["x"]), norm_x = float(np.linalg.norm(s["x"])),
                   Noutliers_board = int(s["Noutliers_board"]),
                   outlier_mask_packed = np.packbits((o["observations_board"][...,2] < 0).ravel()),
                   outlier_mask_shape  = np.array(o["observations_board"].shape[:-1]),
                   # every 997th residual: a spot check of x itself without its 51 MB
                   x_every_997th = s["x"][::997].copy(), Nmeasurements = int(s["x"].size))
        if ref._last_triangulated_flags is not None:
            rec["Noutliers_triangulated_point"] = int(s["Noutliers_triangulated_point"])
            rec["triangulated_flags"] = np.array(ref._last_triangulated_flags).copy()
        for k in ("intrinsics", "rt_cam_ref", "rt_ref_frame", "calobject_warp"):
            if o.get(k) is not None and np.size(o[k]): rec["solved_" + k] = np.array(o[k])
        out = os.path.join(ROOT, "tests", "golden", f"reference_solve_{name}.npz")
        np.savez_compressed(out, **rec)
        print(f"{name}: {dt:.1f} s, rms {s['rms_reproj_error__pixels']:.9f}, {s['Noutliers_board']} board outliers, "
              f"inputs {h[:16]} -> {out} ({os.path.getsize(out)} bytes)", flush=True)
