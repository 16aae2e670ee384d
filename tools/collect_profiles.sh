#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh r02b'): the bench line, and the
# rocprofv3 --kernel-trace --stats tables of the same command and of configuration 2, under gpurun_out/
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns -- python $R/bench.py --no-cpu-baseline --no-full-solve > /dev/null 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_ns "round 2 ($tag), 8 cameras x 1000 frames OPENCV8: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-full-solve (5 warmup + 50 timed steps)" > $R/gpurun_out/${tag}_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/tools/probe_config2.py > $R/gpurun_out/${tag}_config2.log 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_c2 "round 2 ($tag), configuration 2 (1 camera x 800 frames, SPLINED 30x20, core locked): rocprofv3 --kernel-trace --stats -- python tools/probe_config2.py (12 trial steps + one full solve)" > $R/gpurun_out/${tag}_kernel_stats_config2_splined.txt
grep -h "config2\|solve s" $R/gpurun_out/${tag}_config2.log
tail -c 600 $R/gpurun_out/${tag}_bench.json
