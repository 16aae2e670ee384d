#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
counter_collection CSVs -> per-kernel MFMA utilisation (dev tool; the JSON goes to profiles/).

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * NSIMD)
      the fraction of SIMD-cycles of the launch during which a SIMD's matrix pipe was busy
      (SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the SIMDs; GRBM_GUI_ACTIVE = the launch's
       clocks summed over the 8 XCDs; MI355X: 256 CUs x 4 SIMDs)
usage: mfma_util.py <dir> <out.json> kernel-substring [kernel-substring ...]"""
import sys, csv, glob, json, collections
NSIMD = 256*4
d, out, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        name = row.get("Kernel_Name", "")
        for s in subs:
            if s in name:
                acc[s][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for s, cs in acc.items():
    m = {c: sum(v)/len(v) for c, v in cs.items()}
    n = min(len(v) for v in cs.values())
    r = dict(launches=n, counters_mean_per_launch=m)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        r["kernel_clocks"]  = m["GRBM_GUI_ACTIVE"]/8.0
        r["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"]/(m["GRBM_GUI_ACTIVE"]/8.0*NSIMD)
    res[s] = r
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
