# round 6, call 28: the planned rows' finalize by a wave a destination
O=gpurun_out
timeout 1800 python -m pytest tests/test_triangulated.py tests/test_full_size.py tests/test_fuzz_parity.py tests/test_callback_parity.py tests/test_parallel_gpu.py tests/test_factorization_project.py -q -m gpu -x > $O/r06ac_tests.txt 2>&1
for c in 4 5; do for i in 1 2; do timeout 300 python bench.py --only-config $c > $O/r06ac_config${c}_$i.json 2>/dev/null; done; done
