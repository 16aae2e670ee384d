# round 6, call 8: range-aware duals in the triangulated kernel; the tightened seeding test; the sharded outliers against the reference's record
O=gpurun_out
python -m pytest tests/test_triangulated.py tests/test_seeding.py -q -m gpu -x -s > $O/r06h_tests_a.txt 2>&1
python -m pytest tests/test_parallel_gpu.py tests/test_full_size.py -q -m gpu -x -s -k "references_record or recorded or points_and_pairs" > $O/r06h_tests_b.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 5 4; do
    rm -rf /tmp/prof_c$c
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -- python $R/bench.py --only-config $c > $R/$O/r06h_config$c.json 2> /dev/null
    python $R/tools/kernel_stats_table.py /tmp/prof_c$c "round 6 (call 8), configuration $c: rocprofv3 --kernel-trace --stats -- python bench.py --only-config $c" > $R/$O/r06h_kernel_stats_config$c.txt
done
