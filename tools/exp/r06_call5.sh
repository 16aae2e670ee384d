# round 6, call 5: the rest of the suite (from test_full_size on), the NS kernel table, the bench (D2H pipeline, zeroing by 16-byte stores)
O=gpurun_out
timeout 2000 python -m pytest tests -q -m gpu > $O/r06e_gpu_suite.txt 2>&1
python bench.py > $O/r06e_bench.json 2> $O/r06e_bench.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_ns
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns -- python $R/bench.py --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_ns "round 6 (call 5), 8 cameras x 1000 frames OPENCV8: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-full-solve --no-configs" > $R/$O/r06e_kernel_stats.txt
python $R/tools/step_trace_dump.py /tmp/prof_ns 41 > $R/$O/r06e_ns_step_in_time_order.txt 2>&1
for c in 2; do
    rm -rf /tmp/prof_c$c
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -- python $R/bench.py --only-config $c > $R/$O/r06e_config$c.json 2> /dev/null
    python $R/tools/kernel_stats_table.py /tmp/prof_c$c "round 6 (call 5), configuration $c: rocprofv3 --kernel-trace --stats -- python bench.py --only-config $c" > $R/$O/r06e_kernel_stats_config$c.txt
    python $R/tools/step_trace_dump.py /tmp/prof_c$c 8 > $R/$O/r06e_config${c}_step_in_time_order.txt 2>&1
done
