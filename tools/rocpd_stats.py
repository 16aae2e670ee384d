#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration, share) from a
rocprofv3 rocpd sqlite database, i.e. what `rocprofv3 --kernel-trace --stats`
collected. Usage: rocpd_stats.py results.db [> profiles/summary.txt]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select * from kernels").fetchall()
iname = cols.index("name"); istart = cols.index("start"); iend = cols.index("end")
stats = {}
for r in rows:
    d = r[iend] - r[istart]
    s = stats.setdefault(r[iname], [0, 0, 1<<62, 0])
    s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
total = sum(s[1] for s in stats.values())
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for name, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    short = name if len(name) <= 90 else name[:87] + "..."
    print(f"{short:90s} {s[0]:6d} {s[1]/1e3:11.1f} {s[1]/s[0]/1e3:9.2f} {s[2]/1e3:9.2f} {s[3]/1e3:9.2f} {100*s[1]/total:6.2f}")
print(f"{'TOTAL':90s} {sum(s[0] for s in stats.values()):6d} {total/1e3:11.1f}")
