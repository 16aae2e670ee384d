# round 6, call 32: the line once more (call 31's box had a slow host: its suite took 8:41 instead of 5:15)
for i in 1 2; do python bench.py > gpurun_out/r06ag_bench_$i.json 2> gpurun_out/r06ag_bench_$i.err; done
