# round 6, call 15: the triangulated pairs in the board kernel's launch (board_tri_kernel)
O=gpurun_out
timeout 1200 python -m pytest tests/test_triangulated.py tests/test_full_size.py tests/test_fuzz_parity.py -q -m gpu --durations=8 > $O/r06o_tests.txt 2>&1
for i in 1 2; do timeout 300 python bench.py --only-config 5 > $O/r06o_config5_$i.json 2>/dev/null; done
timeout 300 python bench.py --only-config 4 > $O/r06o_config4.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --only-config 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats_table.py /tmp/prof_c5 "round 6 (r06o), BASELINE.json configuration 5: rocprofv3 --kernel-trace --stats -- python bench.py --only-config 5" > $GRAFT_REPO_ROOT/$O/r06o_kernel_stats_config5.txt
