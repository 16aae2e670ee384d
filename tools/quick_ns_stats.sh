#!/bin/bash
# Runs on the GPU box: the NS kernel table (rocprofv3 --kernel-trace --stats of bench.py without the CPU legs), top rows
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- python $R/bench.py --no-cpu-baseline --no-full-solve > /tmp/bench_q.json 2>/dev/null
python $R/tools/kernel_stats_table.py /tmp/prof_q "quick" | cut -c1-60,100-170 | head -${1:-12}
python -c "import json; d=json.loads(open('/tmp/bench_q.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
