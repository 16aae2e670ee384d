# round 6, call 10: the triangulated kernel by two lanes a pair, A/B against the one-lane Dual<12> library on one box
O=gpurun_out
for rep in 1 2; do
  for lib in libmrcal_amd_oldtri.so libmrcal_amd.so; do
    for c in 5 4; do
    MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --only-config $c 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin)[0]; print('$lib config $c', j.get('ms_per_step'), j.get('full_solve',{}).get('seconds'), j.get('error'))" >> $O/r06j_ab_tri.txt
    done
  done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $R/bench.py --only-config 5 > /dev/null 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_c5 "round 6 (call 10), configuration 5" | head -8 > $R/$O/r06j_kernel_stats_config5.txt
cd $R
python -m pytest tests/test_triangulated.py tests/test_full_size.py -q -m gpu -x -k "triang or points_and_pairs or recorded" > $O/r06j_tests.txt 2>&1
