# on the GPU box: configuration 2's kernel table (rocprofv3 --kernel-trace --stats of tools/probe_config2.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/tools/probe_config2.py > $O/c2.log 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_c2 "configuration 2: rocprofv3 --kernel-trace --stats -- python tools/probe_config2.py" > $O/c2_kernel_stats.txt
head -22 $O/c2_kernel_stats.txt | cut -c1-60,100-170
grep config2 $O/c2.log
python $R/tools/step_trace_dump.py /tmp/prof_c2 8 > $O/c2_step_trace.txt 2>&1; cat $O/c2_step_trace.txt | cut -c1-110
