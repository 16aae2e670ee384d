# round 6, call 29: the whole GPU suite and the line + kernel tables at the final code
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > $O/r06ad_gpu_suite.txt 2>&1
bash tools/collect_r06.sh r06ad "bench stats" > $O/r06ad_collect.log 2>&1
