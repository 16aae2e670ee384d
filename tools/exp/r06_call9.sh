# round 6, call 9: A/B of the triangulated kernel (Dual<12> against range-aware duals) on one box; the two tests again
O=gpurun_out
for rep in 1 2; do
  for lib in libmrcal_amd_oldtri.so libmrcal_amd.so; do
    MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --only-config 5 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin)[0]; print('$lib config 5', j.get('ms_per_step'), j.get('full_solve',{}).get('seconds'), j.get('error'))" >> $O/r06i_ab_tri.txt
    MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --only-config 4 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin)[0]; print('$lib config 4', j.get('ms_per_step'), j.get('full_solve',{}).get('seconds'), j.get('error'))" >> $O/r06i_ab_tri.txt
  done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $R/bench.py --only-config 5 > /dev/null 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_c5 "round 6 (call 9), configuration 5" | head -8 > $R/$O/r06i_kernel_stats_config5.txt
cd $R
python -m pytest tests/test_seeding.py tests/test_parallel_gpu.py tests/test_triangulated.py -q -m gpu -x -s -k "real_data or references_record or triang" > $O/r06i_tests.txt 2>&1
