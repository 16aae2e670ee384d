# round 6, call 47: the whole GPU suite at the final code; the line, the kernel tables, the driver's arguments
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > $O/r06av_gpu_suite.txt 2>&1
bash tools/collect_r06.sh r06av "bench stats" > $O/r06av_collect.log 2>&1
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06av_bench_driver_$i.json 2>/dev/null; done
