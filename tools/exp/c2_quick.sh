R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/tools/probe_config2.py > $O/c2.log 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_c2 "configuration 2: rocprofv3 --kernel-trace --stats -- python tools/probe_config2.py" > $O/c2_kernel_stats.txt
head -12 $O/c2_kernel_stats.txt | cut -c1-60,100-170
python $R/tools/step_trace_dump.py /tmp/prof_c2 8 > $O/c2_step_trace.txt 2>&1; cat $O/c2_step_trace.txt | cut -c1-110
cd $R; for i in 1 2; do python tools/probe_config2.py 2>&1 | grep "config2 ms" | cut -c1-40; done
timeout 600 python -m pytest tests/test_solver_parity.py tests/test_callback_parity.py -x -q -m gpu -k "splined" 2>&1 | tail -2
