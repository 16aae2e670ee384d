#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats (csv output) -> the text table kept under profiles/.
usage: kernel_stats_table.py <dir with *_kernel_stats.csv> "<header line>" > profiles/xxx.txt"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
print("#", sys.argv[2])
print(f"{'kernel':100s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
tot_calls, tot = 0, 0.0
for r in csv.DictReader(open(f)):
    n = r["Name"] if len(r["Name"]) <= 100 else r["Name"][:97] + "..."
    print(f"{n:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:11.1f} {float(r['AverageNs'])/1e3:9.2f} "
          f"{float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
    tot_calls += int(r["Calls"]); tot += float(r["TotalDurationNs"])/1e3
print(f"{'TOTAL':100s} {tot_calls:6d} {tot:11.1f}")
