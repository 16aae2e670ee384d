# round 6, call 11: the whole GPU suite at the final code, then everything profiles/r06_* is made of
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > $O/r06_gpu_suite.txt 2>&1
bash tools/collect_r06.sh r06 > $O/r06_collect.log 2>&1
