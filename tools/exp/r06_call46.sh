# round 6, call 46: the first diagonal block's chain starts when ITS rows have arrived: stamps, A/B, tests
O=gpurun_out
MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs 2>&1 | grep "chol ts" | tail -2 > $O/r06au_chol_ts.txt
rm -f $O/r06au_ab.txt
for rep in 1 2 3; do
  for lib in libmrcal_amd_prev.so libmrcal_amd.so; do
    MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-solve --no-configs 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib ns --steps 20 --warmup 5', j.get('ms_per_step'), j.get('value'))" >> $O/r06au_ab.txt
    MRCAL_AMD_LIB=mrcal_amd/$lib python bench.py --only-config 1 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin)[0]; print('$lib config 1', j.get('ms_per_step'), j.get('error'))" >> $O/r06au_ab.txt
  done
done
timeout 2400 python -m pytest tests/test_solver_parity.py tests/test_full_size.py tests/test_graph_mode.py tests/test_triangulated.py tests/test_moving_camera.py tests/test_parallel_gpu.py -q -m gpu -x > $O/r06au_tests.txt 2>&1
