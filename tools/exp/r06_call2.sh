# round 6, call 2: where do the 12 us between a wave's life and the kernel's time go at 1600 observations? + the J-free tests
O=gpurun_out
for c in 1 5 ns; do
    MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so python tools/probe_board_ts.py $c > $O/r06b_board_ts_$c.txt 2>&1
done
python -m pytest tests/test_solver_parity.py -x -q -m gpu -k "jacobian_stream" > $O/r06b_jfree_small.txt 2>&1
python -m pytest tests/test_full_size.py -x -q -m gpu -s -k "jacobian_stream" > $O/r06b_jfree_full.txt 2>&1
