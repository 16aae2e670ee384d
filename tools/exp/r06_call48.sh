# round 6, call 48: how long a tile of the first panel's trailing update takes in a worker wave (-DCHOL_TS)
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts wave" | tail -12 > gpurun_out/r06aw_chol_tiles.txt
