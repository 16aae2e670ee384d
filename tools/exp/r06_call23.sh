# round 6, call 23: the whole GPU suite at the final code, then everything profiles/r06_* is made of
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > $O/r06w_gpu_suite.txt 2>&1
( time python bench.py > /dev/null 2> /dev/null ) 2> $O/r06w_bench_wall.txt
bash tools/collect_r06.sh r06w > $O/r06w_collect.log 2>&1
