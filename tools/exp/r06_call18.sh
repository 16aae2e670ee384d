# round 6, call 18: the back-substitution in the factorization's launch: the suite, then A/B on one box
O=gpurun_out
timeout 900 python -m pytest tests/test_solver_parity.py -q -m gpu -x -k "backsubstitution_in_the" > $O/r06r_backsub_test.txt 2>&1
timeout 600 python tools/exp/r06_ab_backsub.py > $O/r06r_ab_backsub_ns.txt 2>&1
timeout 600 python tools/exp/r06_ab_backsub.py --only-config 1 > $O/r06r_ab_backsub_c1.txt 2>&1
timeout 600 python tools/exp/r06_ab_backsub.py --only-config 5 > $O/r06r_ab_backsub_c5.txt 2>&1
for c in 4 5; do timeout 300 python tools/probe_oneshot.py $c > $O/r06r_oneshot_$c.txt 2>&1; done
timeout 2400 python -m pytest tests -q -m gpu -x > $O/r06r_gpu_suite.txt 2>&1
