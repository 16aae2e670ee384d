# round 6, call 43: the right-hand side as row Nc of the packed copy: stamps, the whole GPU suite, the driver's bench
O=gpurun_out
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts" | tail -2 > $O/r06ar_chol_ts.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/r06ar_gpu_suite.txt 2>&1
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06ar_bench_driver_$i.json 2>/dev/null; done
