#!/bin/bash
# per-kernel times of the solver step under rocprofv3 (dev tool). usage: probe_step_kernels.sh <tag> [env assignments...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-solve > $GRAFT_REPO_ROOT/gpurun_out/bench_$tag.log 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats_table.py $GRAFT_REPO_ROOT/gpurun_out/prof_$tag "$tag" | cut -c1-50,100-150 | head -${NLINES:-8}
tail -1 $GRAFT_REPO_ROOT/gpurun_out/bench_$tag.log | cut -c1-200
