#!/usr/bin/env python3
"""Makes tests/golden/real_*.cameramodel: real calibrations (detected chessboard
corners, outliers and all) with their optimization_inputs, so that stored solves
can be replayed through the GPU path where /root/reference does not exist.

Source: the calibration of the reference's own documentation,
doc/data/figueroa-overpass-looking-S/{opencv8,splined}-0.cameramodel (1 camera,
186 frames of a 10x10 board, 6016x4016 imager). The files are read with
mrcal_amd.cameramodel and written back with its writer: what is committed is
this repo's rendering of the same data (the base-85 blob is re-encoded from the
decoded dict; the comment header of the source is not carried over).

    python3 tests/golden/make_real_calibration.py
"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
from mrcal_amd.cameramodel import cameramodel

SRC = "/root/reference/doc/data/figueroa-overpass-looking-S"
for name in ("opencv8-0", "splined-0"):
    m = cameramodel(os.path.join(SRC, name + ".cameramodel"))
    oi = m.optimization_inputs()
    m2 = cameramodel(optimization_inputs=oi, icam_intrinsics=m.icam_intrinsics())
    m2.valid_intrinsics_region(m.valid_intrinsics_region())
    out = os.path.join(HERE, f"real_{name}.cameramodel")
    m2.write(out, note=f"real calibration data: the reference's doc/data/figueroa-overpass-looking-S/{name}.cameramodel,\n"
                       "re-written by tests/golden/make_real_calibration.py")
    # what was written is what was read
    m3 = cameramodel(out)
    a, b = m.optimization_inputs(), m3.optimization_inputs()
    assert sorted(a.keys()) == sorted(b.keys()), (sorted(a.keys()), sorted(b.keys()))
    for k in a:
        if isinstance(a[k], np.ndarray): assert np.array_equal(a[k], b[k]), k
        else:                            assert a[k] == b[k], k
    assert np.array_equal(m.intrinsics()[1], m3.intrinsics()[1]) and m.intrinsics()[0] == m3.intrinsics()[0]
    print(out, os.path.getsize(out), "bytes;", oi["observations_board"].shape, "observations")
