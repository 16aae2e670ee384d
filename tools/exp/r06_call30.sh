# round 6, call 30: the quadratic form's workgroups in the factorization's launch: A/B of two builds on one box, then the solver's tests
O=gpurun_out
rm -f $O/r06ae_ab_quadform.txt
for rep in 1 2 3; do
  for lib in libmrcal_amd_prev.so libmrcal_amd.so; do
    MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --no-cpu-baseline --no-full-solve --no-configs 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib ns', j.get('ms_per_step'), j.get('ms_per_step_no_jacobian_stream'))" >> $O/r06ae_ab_quadform.txt
    MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --only-config 1 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin)[0]; print('$lib config 1', j.get('ms_per_step'), j.get('error'))" >> $O/r06ae_ab_quadform.txt
  done
done
timeout 1800 python -m pytest tests/test_solver_parity.py tests/test_full_size.py tests/test_graph_mode.py tests/test_triangulated.py -q -m gpu -x > $O/r06ae_tests.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ns
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_stats_table.py /tmp/prof_ns "round 6 (r06ae): the metric's problem" | head -12 > $GRAFT_REPO_ROOT/$O/r06ae_kernel_stats.txt
