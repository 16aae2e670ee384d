MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python tools/probe_config2.py > gpurun_out/lch_ts.log 2>&1
grep "lchol" gpurun_out/lch_ts.log | sort | uniq -c | sort -rn | head -3
grep "lchol" gpurun_out/lch_ts.log | tail -12
