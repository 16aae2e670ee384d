# round 6, call 36: the one-workgroup Cholesky's phases (-DCHOL_TS) at the metric's size
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts" | tail -6 > gpurun_out/r06ak_chol_ts.txt
