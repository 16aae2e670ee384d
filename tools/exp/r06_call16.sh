# round 6, call 16: where a one-shot optimize() of a structure-from-motion problem spends its time; the merged launch's counters; the suite
O=gpurun_out
for c in 4 5 ns; do timeout 300 python tools/probe_oneshot.py $c > $O/r06p_oneshot_$c.txt 2>&1; done
bash tools/collect_r06_pmc.sh r06p "5" > /dev/null 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > $O/r06p_gpu_suite.txt 2>&1
