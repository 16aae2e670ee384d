#!/bin/bash
# Runs on the GPU box: per-kernel time of ONE rank's trial step at N = 2, 4, 8 ranks (tools/probe_shard_compute.py under
# rocprofv3 --kernel-trace --stats), the step kernels only, averaged per call -> markdown rows for DESIGN.md section 7
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in 2 4 8; do
    rm -rf /tmp/prof_s$n
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s$n -- python $R/tools/probe_shard_compute.py $n "$@" > /dev/null 2>&1
    python - $n /tmp/prof_s$n <<'PY'
import csv, glob, sys
n, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
want = ("board_prologue_kernel<true>", "board_kernel<0, 8, true, true, true>", "assemble_factor_kernel", "schur_syrk_mfma_kernel",
        "step2_reduce_kernel", "schur_cholesky_solve_kernel", "step2_backsub_quadform_kernel", "step2_pack2_kernel", "lchol_", "schur_syrk_strip_kernel",
        "step2_finish_kernel", "step2_post_kernel")
rows, total = [], 0.0
steps = None
for r in csv.DictReader(open(f)):
    name = r["Name"]
    if not any(w in name for w in want): continue
    calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])/1e3
    import re
    short = re.search(r"mrcal_amd::(\w+)", name).group(1)
    if "assemble_factor_kernel" in name: steps = calls
    rows.append((short, calls, tot))
steps = steps or max(c for _, c, _ in rows)
per = {}
for short, calls, tot in rows: per[short] = per.get(short, 0.0) + tot/steps
print(f"N={n}: " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1])) + f"  | sum {sum(per.values()):.1f} us per step")
PY
done
