#!/usr/bin/env python3
"""Per-kernel durations and the GAPS between consecutive kernels of the trial step, from a rocprofv3 --kernel-trace
CSV directory (dev tool): python tools/step_timeline.py <dir>"""
import sys, csv, glob, collections
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    for k in ("step2_choose", "board_prologue", "board_kernel", "assemble_factor", "schur_syrk", "step2_reduce", "schur_cholesky", "step2_backsub"):
        if k in n: return k
    return None
seq = [(s, e, short(n)) for s, e, n in rows if short(n)]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(seq[:-1], seq[1:]):
    dur[n0].append(e0 - s0)
    gap[n0 + " -> " + n1].append(s1 - e0)
print("durations (us):")
for k, v in dur.items(): v = sorted(v); print(f"  {k:20s} n={len(v):4d} median {v[len(v)//2]/1e3:7.2f}")
print("gaps (us):")
tot = 0
for k, v in gap.items():
    v = sorted(v)
    if len(v) < 20: continue
    print(f"  {k:40s} n={len(v):4d} median {v[len(v)//2]/1e3:6.2f}"); tot += v[len(v)//2]/1e3
print("sum of median gaps", tot)
