#!/bin/bash
# configuration 2 with the dissection: one factoring step in time order, every launch of the factorization
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_nd
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nd -- python $R/tools/probe_config2.py > /dev/null 2>&1
python $R/tools/step_trace_dump.py /tmp/prof_nd 8 | grep -v lchol | tee $O/nd_step.txt
python $R/tools/exp/lchol_launches.py /tmp/prof_nd 8 | tee $O/nd_launches.txt
