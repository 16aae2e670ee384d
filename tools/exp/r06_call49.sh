# round 6, call 49: the trailing update's tile with its LDS reads asked for together: stamps
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts" | tail -2 > gpurun_out/r06ax_chol_ts.txt
