# round 6, call 22: the planned rows in assemble_factor_kernel's launch
O=gpurun_out
timeout 1500 python -m pytest tests/test_triangulated.py tests/test_full_size.py tests/test_fuzz_parity.py tests/test_callback_parity.py tests/test_parallel_gpu.py -q -m gpu -x > $O/r06v_tests.txt 2>&1
for i in 1 2; do timeout 300 python bench.py --only-config 5 > $O/r06v_config5_$i.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --only-config 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats_table.py /tmp/prof_c5 "round 6 (r06v), BASELINE.json configuration 5: rocprofv3 --kernel-trace --stats -- python bench.py --only-config 5" > $GRAFT_REPO_ROOT/$O/r06v_kernel_stats_config5.txt
