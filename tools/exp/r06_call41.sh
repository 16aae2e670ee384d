# round 6, call 41: a head start for the factorization's load in front of the quadratic form's workgroups: stamps
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts" | tail -3 > gpurun_out/r06ap_chol_ts.txt
