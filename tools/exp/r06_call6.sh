# round 6, call 6: after the cut of solver_kernels.hip into translation units and the switch prune: the whole suite + bench
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > $O/r06f_gpu_suite.txt 2>&1
python bench.py > $O/r06f_bench.json 2> $O/r06f_bench.err
