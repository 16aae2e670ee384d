#!/usr/bin/env python3
"""every launch of the big camera block's factorization in ONE trial step, in time order (rocprofv3 --kernel-trace CSV dir, which step)"""
import sys, csv, glob
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mrcal_amd::", "").replace("void ", ""), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "board_prologue_kernel<true>" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts)//2
i0, i1 = starts[k], starts[k+1]
t0 = rows[i0][0]; prev = t0
for s, e, n, g, w in rows[i0:i1]:
    if "lchol" in n: print(f"{(s-t0)/1e3:9.2f} +{(e-s)/1e3:7.2f} gap {(s-prev)/1e3:6.2f}  {n[:28]:28s} grid {g} / {w}")
    prev = max(prev, e)
