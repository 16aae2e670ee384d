# round 6, call 28: and by two streams
O=gpurun_out
cat /sys/kernel/mm/transparent_hugepage/enabled > $O/r06ab_thp.txt 2>&1
for i in 1 2; do python bench.py --no-cpu-baseline --no-configs > $O/r06ab_bench_$i.json 2> /dev/null; done
timeout 300 python tools/probe_oneshot.py ns > $O/r06ab_oneshot.txt 2>&1
timeout 900 python -m pytest tests/test_callback_parity.py tests/test_operator_boundary.py -q -m gpu -x > $O/r06ab_tests.txt 2>&1
