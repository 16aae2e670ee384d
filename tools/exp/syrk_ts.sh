MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python tools/probe_config2.py > gpurun_out/syrk_ts.log 2>&1
grep "syrk strip" gpurun_out/syrk_ts.log | tail -8
bash tools/exp/c2_quick.sh 2>&1 | grep -v "^lchol\|^W2026"
