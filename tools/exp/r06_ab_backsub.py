#!/usr/bin/env python3
"""bench.py's line (steps only) with the back-substitution in the factorization's launch (the default) and as a launch of
its own (the test hook separate_backsub), alternating in ONE process-per-run on one box (dev tool)
usage: r06_ab_backsub.py [bench.py arguments]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:] or ["--no-cpu-baseline", "--no-full-solve", "--no-configs"]
code = "import sys, runpy; sys.path.insert(0, %r); import mrcal_amd; mrcal_amd.set_test_hook('separate_backsub', int(sys.argv[1])); sys.argv = ['bench.py'] + sys.argv[2:]; runpy.run_path(%r, run_name='__main__')" % (ROOT, os.path.join(ROOT, "bench.py"))
for rep in range(3):
    for sep in (1, 0):
        r = subprocess.run([sys.executable, "-c", code, str(sep)] + args, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{") or l.startswith("[")]
        if not line: print("separate_backsub", sep, "FAILED", r.stderr[-500:]); continue
        d = json.loads(line[-1])
        for c in (d if isinstance(d, list) else [d]):
            print("separate_backsub", sep, c.get("config", {}).get("workload", "")[:40] if isinstance(c.get("config"), dict) else "config " + str(c.get("config")), "ms_per_step", c["ms_per_step"], flush=True)
