#!/usr/bin/env python3
"""gpurun_out/<tag>_* (what tools/collect_r06.sh left on the GPU box) -> profiles/r06_* (tracked): copies what is a table
already, formats the counters' raw summaries (per launch, per wave), fills the two HBM-traffic files bench.py reads.
usage: python tools/publish_r06.py [tag]         (dev tool; run in the build container after the gpurun call)"""
import sys, os, re, json, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, Pf = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"

def copy(src, dst):
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copyfile(s, os.path.join(Pf, dst)); print("  ", dst)
    else:
        print("   (missing:", src, ")")

# the tables
bj = os.path.join(G, f"{tag}_bench.json")
if os.path.exists(bj):
    line = [l for l in open(bj).read().splitlines() if l.startswith("{")][-1]
    open(os.path.join(Pf, "r06_bench.json"), "w").write(line + "\n"); print("   r06_bench.json")
for src, dst in (("kernel_stats.txt", "r06_kernel_stats.txt"), ("kernel_stats_config1.txt", "r06_kernel_stats_config1.txt"),
                 ("kernel_stats_config2.txt", "r06_kernel_stats_config2_splined.txt"), ("kernel_stats_config3.txt", "r06_kernel_stats_config3.txt"),
                 ("kernel_stats_config5.txt", "r06_kernel_stats_config5.txt"), ("ns_step_in_time_order.txt", "r06_ns_step_in_time_order.txt"),
                 ("config2_step_in_time_order.txt", "r06_config2_step_in_time_order.txt"), ("config2_lchol_launches.txt", "r06_config2_lchol_launches.txt"),
                 ("dispatch_rate.txt", "r06_dispatch_rate.txt"), ("handoff_32k.txt", "r06_handoff_32k.txt"), ("board_ts_config1.txt", "r06_board_ts_config1.txt"),
                 ("rccl_world1_measured.json", "r06_rccl_world1_measured.json"), ("config3_solve_vs_reference.json", "r06_config3_solve_vs_reference.json"),
                 ("config5_solve_vs_reference.json", "r06_config5_solve_vs_reference.json"),
                 ("ns_sharded_outliers_vs_reference.json", "r06_ns_sharded_outliers_vs_reference.json")):
    copy(f"{tag}_{src}" if os.path.exists(os.path.join(G, f"{tag}_{src}")) else "r06_" + src, dst)     # (the records and the RCCL worker write their own r06_ names)

# the counters: raw summary (tools/pmc_summary.py) -> per launch and per wave
def parse_raw(path):
    kernels, cur = {}, None
    for l in open(path):
        if l.startswith("#") or l.startswith("W2") or not l.strip(): continue
        m = re.match(r"\s+(\w+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", l)
        if m and cur is not None: kernels[cur][m.group(1)] = float(m.group(3))
        elif not l.startswith(" "): cur = l.strip(); kernels[cur] = {}
    return kernels
WHAT = {"ns": "the metric's problem (8 cameras x 1000 frames OPENCV8: 8000 observations)", "1": "BASELINE configuration 1 (1600 observations, OPENCV8)",
        "2": "BASELINE configuration 2 (800 observations of 10x10 corners, splined 30x20)", "5": "BASELINE configuration 5 (1600 board observations, OPENCV4, beside 67 000 triangulated pairs)"}
traffic = {}
for c in ("ns", "1", "2", "5"):
    raw = os.path.join(G, f"{tag}_config{c}_jacobian_kernel_pmc_raw.txt")
    if not os.path.exists(raw): print("   (missing:", raw, ")"); continue
    ks = parse_raw(raw)
    out = [f"# The Jacobian kernel(s) of {WHAT[c]}: rocprofv3 --kernel-trace --pmc <4 counters a pass> -- python tools/probe_board_one.py 1 0 5 {c}",
           "# (tools/collect_r06_pmc.sh: six passes + WRITE_SIZE and FETCH_SIZE a pass each; means over the launches; SQ cycle counters count in units of 4 clocks;",
           "#  WRITE_SIZE / FETCH_SIZE in KiB, FETCH_SIZE to be doubled: MI355X_MICROARCH.md). Round 6, the code as committed"]
    for name, cs in ks.items():
        waves = cs.get("SQ_WAVES", 0.0)
        # the variant that is launched with the Jacobian: the others are the x-only launches of the probe's set-up
        out.append(f"## {name}   ({int(waves)} waves a launch)")
        out.append(f"# {'counter':32s}{'per launch':>16s}{'per wave':>14s}")
        for k in sorted(cs):
            out.append(f"{k:34s}{cs[k]:16.1f}" + (f"{cs[k]/waves:14.1f}" if waves > 0 and k not in ("WRITE_SIZE", "FETCH_SIZE", "GRBM_GUI_ACTIVE") else ""))
        if waves > 0 and "SQ_WAVE_CYCLES" in cs:
            life = 4*cs["SQ_WAVE_CYCLES"]/waves
            out.append(f"# a wave lives {life:.0f} clocks; issuing {100*cs.get('SQ_ACTIVE_INST_ANY', 0)/cs['SQ_WAVE_CYCLES']:.0f} %, "
                       f"waiting at issue {100*cs.get('SQ_WAIT_INST_ANY', 0)/cs['SQ_WAVE_CYCLES']:.0f} %, waiting for anything {100*cs.get('SQ_WAIT_ANY', 0)/cs['SQ_WAVE_CYCLES']:.0f} %; "
                       f"the launch {cs.get('GRBM_GUI_ACTIVE', 0)/8:.0f} clocks")
        if "WRITE_SIZE" in cs and "FETCH_SIZE" in cs and (cs["WRITE_SIZE"] > 64 or "true" in name or "rows" in name):
            traffic.setdefault(c, []).append((name, cs["WRITE_SIZE"], cs["FETCH_SIZE"]))
        out.append("")
    dst = "r06_board_kernel_pmc.txt" if c == "ns" else f"r06_config{c}_jacobian_kernel_pmc.txt"
    open(os.path.join(Pf, dst), "w").write("\n".join(out)); print("  ", dst)

# HBM traffic: the launch that writes the Jacobian (the largest writer of the configuration's kernels)
def biggest(c):
    return max(traffic[c], key=lambda e: e[1]) if c in traffic else None
tp = os.path.join(Pf, "r06_jacobian_kernel_hbm_traffic.json")
tj = json.load(open(tp))
for c in ("1", "2", "5"):
    b = biggest(c)
    if b is None: continue
    name, w, f = b
    tj["configs"][c] = dict(kernel=name, write_bytes=int(round(w*1024)), fetch_bytes=int(round(2*f*1024)), hbm_bytes_per_launch=int(round(w*1024 + 2*f*1024)),
                            raw=dict(WRITE_SIZE_KiB=w, FETCH_SIZE_KiB=f))
tj["taken"] = f"round 6 (tools/collect_r06.sh pmc, tools/publish_r06.py {tag}): the kernels as committed (configuration 2: board_splined_rows_kernel)"
json.dump(tj, open(tp, "w"), indent=1); print("   r06_jacobian_kernel_hbm_traffic.json")
b = biggest("ns")
if b is not None:
    bp = os.path.join(Pf, "board_kernel_hbm_traffic.json")
    bjs = json.load(open(bp))
    name, w, f = b
    bjs["raw"]["round5_WRITE_SIZE_KiB"], bjs["raw"]["round5_FETCH_SIZE_KiB"] = bjs["raw"]["WRITE_SIZE_KiB"], bjs["raw"]["FETCH_SIZE_KiB"]
    bjs["raw"]["WRITE_SIZE_KiB"], bjs["raw"]["FETCH_SIZE_KiB"] = w, f
    bjs["write_bytes"], bjs["fetch_bytes"] = int(round(w*1024)), int(round(2*f*1024))
    bjs["hbm_bytes_per_launch"] = float(bjs["write_bytes"] + bjs["fetch_bytes"])
    bjs["taken"] = f"round 6 (tools/collect_r06.sh pmc, tools/publish_r06.py {tag})"
    json.dump(bjs, open(bp, "w"), indent=2); print("   board_kernel_hbm_traffic.json")

# matrix-pipe utilisation
mn, m2 = os.path.join(G, f"{tag}_mfma_ns.json"), os.path.join(G, f"{tag}_mfma_config2.json")
if os.path.exists(mn) and os.path.exists(m2):
    r5 = json.load(open(os.path.join(Pf, "r05_mfma_utilisation.json")))
    out = dict(workload_cameras=8, workload_frames=1000,
               how="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE (counters in a pass of their own) around "
                   "bench.py --steps 10 --no-cpu-baseline --no-full-solve --no-configs (ns) and bench.py --only-config 2 (config2); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                   "(GRBM_GUI_ACTIVE/8 x 1024 SIMDs) (tools/mfma_util.py). Round 6 (tools/collect_r06.sh mfma)",
               ns=json.load(open(mn)), config2=json.load(open(m2)))
    ll = os.path.join(G, f"{tag}_config2_lchol_launches.txt")
    if os.path.exists(ll):
        rows = [l.split() for l in open(ll) if "lchol" in l]
        t0, t1 = float(rows[0][0]), float(rows[-1][0]) + float(rows[-1][2])
        lch = dict(r5["lchol"]); lch["round5"] = dict(span_us=lch["span_us"], tflops=lch["tflops"], frac_of_fp64_matrix_peak=lch["frac_of_fp64_matrix_peak"])
        lch["span_us"] = t1 - t0; lch["tflops"] = lch["flops"]/(t1 - t0)/1e6; lch["frac_of_fp64_matrix_peak"] = lch["tflops"]/78.6
        lch["what"] = lch["what"].replace("r05_config2", "r06_config2")
        out["lchol"] = lch
    json.dump(out, open(os.path.join(Pf, "r06_mfma_utilisation.json"), "w"), indent=1); print("   r06_mfma_utilisation.json")
