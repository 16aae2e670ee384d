#!/bin/bash
# configuration 2: one factoring step in time order, for each of the given environments; then the step time of each, twice
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  rm -rf /tmp/prof_c2_$i
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2_$i -- python $R/tools/probe_config2.py > /dev/null 2>&1
  echo "== [$v]"; python $R/tools/step_trace_dump.py /tmp/prof_c2_$i 8 | cut -c1-100
  i=$((i+1))
done 2>&1 | tee $O/r05_c2_traces.txt
cd $R
for k in 1 2; do for v in "$@"; do echo "[$v] $(env $v python tools/probe_config2.py 2>&1 | grep 'config2 ms' | cut -c1-44)"; done; done | tee $O/r05_c2_times.txt
