MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python tools/probe_config2.py > gpurun_out/spl_ts.log 2>&1
grep "splined assembly f [0-9]*: wall" gpurun_out/spl_ts.log | tail -14
