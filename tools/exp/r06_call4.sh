# round 6, call 4: pose records by 8 lanes an observation; the splined Jacobian by a lane per row - suite + bench
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r06d_gpu_suite.txt 2>&1
python bench.py > $O/r06d_bench.json 2> $O/r06d_bench.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 2 1; do
    rm -rf /tmp/prof_c$c
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -- python $R/bench.py --only-config $c > $R/$O/r06d_config$c.json 2> /dev/null
    python $R/tools/kernel_stats_table.py /tmp/prof_c$c "round 6 (call 4), configuration $c: rocprofv3 --kernel-trace --stats -- python bench.py --only-config $c" > $R/$O/r06d_kernel_stats_config$c.txt
done
