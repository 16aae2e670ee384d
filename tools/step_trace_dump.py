#!/usr/bin/env python3
"""The kernels of ONE trial step in time order, with start offsets and durations (us), from a rocprofv3
--kernel-trace CSV directory: python tools/step_trace_dump.py <dir> [which_step]   (dev tool)
A step begins at a step2_choose_kernel (or at the prologue launch that carries the choice). Shows what overlaps what when a step uses two streams"""
import sys, csv, glob
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mrcal_amd::", "").replace("void ", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "step2_choose" in r[2]]
if len(starts) < 3:     # the choice rides in the prologue launch
    starts = [i for i, r in enumerate(rows) if "board_prologue_kernel<true>" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts)//2
i0, i1 = starts[k], starts[k+1]
t0 = rows[i0][0]
prev_end = t0
npanel = 0
for s, e, n in rows[i0:i1]:
    if "lchol_panel" in n:
        npanel += 1
        if npanel not in (1, 2, 19): prev_end = max(prev_end, e); continue
    print(f"{(s-t0)/1e3:9.2f} +{(e-s)/1e3:7.2f}  gap {(s-prev_end)/1e3:6.2f}  {n[:60]}")
    prev_end = max(prev_end, e)
print("step", (rows[i1][0]-t0)/1e3, "us")
