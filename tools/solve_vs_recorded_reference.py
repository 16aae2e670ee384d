#!/usr/bin/env python3
"""On the GPU box: BASELINE.json's configurations 3 and 5 SOLVED by mrcal_amd.optimize() and compared with the record
of the reference's own mrcal_optimize() on the same inputs (tests/golden/reference_solve_<name>.npz, made by
tests/golden/make_reference_solves.py in the build container: 13 minutes of one core at configuration 3).
    python tools/solve_vs_recorded_reference.py [outdir] [names...]   ->  <outdir>/r06_<name>_solve_vs_reference.json
The suite asserts the same bounds: tests/test_full_size.py::test_solve_matches_the_references_recorded_solve"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mrcal_amd
from mrcal_amd._cabi import MrcalLib
from mrcal_amd._api import Api
from test_full_size import compare_with_recorded_reference_solve
outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
names = sys.argv[2:] or ["config3", "config5"]
ref = Api(MrcalLib(os.path.join(ROOT, "oracle", "_ref", "libmrcal_ref.so")))
for name in names:
    c = compare_with_recorded_reference_solve(mrcal_amd, ref, name)
    with open(os.path.join(outdir, f"r06_{name}_solve_vs_reference.json"), "w") as f:
        json.dump(c, f, indent=1)
    print(json.dumps(c, indent=1))
