# round 6, call 26: the line and the kernel tables at the final code
bash tools/collect_r06.sh r06z "bench stats" > gpurun_out/r06z_collect.log 2>&1
