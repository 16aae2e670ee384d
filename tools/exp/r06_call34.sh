# round 6, call 34: the quadratic form in the FIRST launch of the large Cholesky (configurations 2 and 3): A/B of two builds, tests
O=gpurun_out
rm -f $O/r06ai_ab_quadform_large.txt
for rep in 1 2 3; do
  for lib in libmrcal_amd_prev.so libmrcal_amd.so; do
    for c in 2 3; do
      MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --only-config $c 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin)[0]; print('$lib config $c', j.get('ms_per_step'), j.get('ms_per_step_no_jacobian_stream'), j.get('full_solve_resident',{}).get('seconds'), j.get('error'))" >> $O/r06ai_ab_quadform_large.txt
    done
  done
done
timeout 2400 python -m pytest tests/test_solver_parity.py tests/test_full_size.py tests/test_splined_subboxes.py tests/test_graph_mode.py -q -m gpu -x > $O/r06ai_tests.txt 2>&1
