# round 6, call 39: the chain in registers alone (the partial panel as it was): stamps, the whole GPU suite
O=gpurun_out
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts" | tail -2 > $O/r06an_chol_ts.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/r06an_gpu_suite.txt 2>&1
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06an_bench_driver_$i.json 2>/dev/null; done
