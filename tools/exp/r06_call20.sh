# round 6, call 20: the same without the fence by every thread
O=gpurun_out
timeout 900 python -m pytest tests/test_solver_parity.py -q -m gpu -x -k "backsubstitution_in_the" > $O/r06t_backsub_test.txt 2>&1
timeout 600 python tools/exp/r06_ab_backsub.py > $O/r06t_ab_backsub_ns.txt 2>&1
timeout 600 python tools/exp/r06_ab_backsub.py --only-config 1 > $O/r06t_ab_backsub_c1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ns
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/step_trace_dump.py /tmp/prof_ns 41 > $GRAFT_REPO_ROOT/$O/r06t_ns_step_in_time_order.txt 2>&1
