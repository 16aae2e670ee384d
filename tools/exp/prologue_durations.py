#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV directory: every board_prologue_kernel<true> launch of the run with its duration, the
duration of the factorization of the step before it (short: that step was rejected / did not factor) and of the board kernel
behind it (dev tool): does the prologue's length depend on what the step before it did?
usage: prologue_durations.py <dir>"""
import sys, csv, glob
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mrcal_amd::", "").replace("void ", "")))
rows.sort()
last_chol = None
out = []
for i, (s, e, n) in enumerate(rows):
    if "cholesky_solve" in n: last_chol = (e - s)/1e3
    if "board_prologue_kernel<true>" in n:
        prev = rows[i-1] if i > 0 else None
        out.append(((e - s)/1e3, last_chol, prev[2][:40] if prev else "", (s - prev[1])/1e3 if prev else 0.0))
print("# prologue us | factorization of the step before, us | the launch in front | gap to it, us")
for d, c, pn, g in out: print(f"{d:8.2f} {c if c is not None else -1:8.2f}  {pn:40s} {g:7.2f}")
import statistics
fact = [d for d, c, _, _ in out if c is not None and c > 20]
nofact = [d for d, c, _, _ in out if c is not None and c <= 20]
if fact and nofact: print(f"# behind a step that factored: n {len(fact)} median {statistics.median(fact):.2f}; behind one that did not: n {len(nofact)} median {statistics.median(nofact):.2f}")
