# round 6, call 17: the drop-in calls of a structure-from-motion problem, piece by piece; the unit cut of kernels.hip under the suite
O=gpurun_out
for c in 4 5; do timeout 300 python tools/exp/r06_sfm_costs.py $c > $O/r06q_sfm_costs_$c.txt 2>&1; done
for c in 4 5; do timeout 300 python tools/probe_oneshot.py $c > $O/r06q_oneshot_$c.txt 2>&1; done
timeout 1500 python -m pytest tests/test_triangulated.py tests/test_solver_parity.py tests/test_full_size.py -q -m gpu -x > $O/r06q_tests.txt 2>&1
