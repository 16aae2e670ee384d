# round 6, call 14: everything profiles/r06_* is made of, at the final code (assemble_splined_kernel without scratch)
O=gpurun_out
bash tools/collect_r06.sh r06n > $O/r06n_collect.log 2>&1
timeout 900 python -m pytest tests/test_solver_parity.py tests/test_full_size.py -q -m gpu -k "splined or config2 or spl" > $O/r06n_splined_tests.txt 2>&1
