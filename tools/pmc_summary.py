#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection CSVs per kernel name (dev tool).
usage: pmc_summary.py dir [kernel-substring]"""
import sys, csv, glob, collections
sub = sys.argv[2] if len(sys.argv) > 2 else "board_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        name = row.get("Kernel_Name", "")
        if sub not in name: continue
        acc[name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
