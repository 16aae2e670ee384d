# round 6, call 21: wall-clock stamps inside step2_chol_backsub_kernel
O=gpurun_out
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs > $O/r06u_cholb_ts.txt 2>&1
