#!/bin/bash
# the metric's step time under two (or more) environments, alternating on ONE box: bash tools/exp/ab_ns_env.sh REPS "ENV_A" "ENV_B" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
reps=$1; shift
cd $R
cat > /tmp/ab_fmt.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
f=lambda v: ('%.4f' % v) if isinstance(v,float) else str(v)
print('ms/step %.4f  it/s %.0f  kernel_ms %.4f frac %.3f | stream frac %s poses %s first-store %s stream %s | solve %.4f s %d it' % (d['ms_per_step'], d['value'], r['kernel_ms_avg'], r['frac'], f(r.get('frac_jacobian_stream')), f(r.get('poses_ms_avg')), f(r.get('first_store_after_ms_avg')), f(r.get('jacobian_stream_ms_avg')), d['full_solve']['seconds'], d['full_solve']['iterations']))
PY
for k in $(seq 1 $reps); do
  for v in "$@"; do
    echo "[$v] $(env $v python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python /tmp/ab_fmt.py)"
  done
done | tee $O/ab_ns_env.txt
