# round 6, call 31: the whole GPU suite at the final code, then everything profiles/r06_* is made of, once more
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > $O/r06af_gpu_suite.txt 2>&1
bash tools/collect_r06.sh r06af > $O/r06af_collect.log 2>&1
