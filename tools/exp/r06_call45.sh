# round 6, call 45: the backward chain's broadcasts by DPP: stamps, the solver's tests
O=gpurun_out
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts" | tail -2 > $O/r06at_chol_ts.txt
timeout 2400 python -m pytest tests/test_solver_parity.py tests/test_full_size.py tests/test_graph_mode.py tests/test_triangulated.py tests/test_moving_camera.py -q -m gpu -x > $O/r06at_tests.txt 2>&1
