# round 6, call 44: the line and the kernel tables at the final code
bash tools/collect_r06.sh r06as "bench stats" > gpurun_out/r06as_collect.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06as_bench_driver.json 2>/dev/null
