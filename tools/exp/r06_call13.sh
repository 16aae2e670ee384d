# round 6, call 13: the fallback that lets the pass stand; the problems torn down behind the caller's back
O=gpurun_out
timeout 900 python -m pytest tests/test_solver_parity.py -q -m gpu -k "disputed_fuzz or explicit_inverse or jacobian_stream" > $O/r06m_fallback_tests.txt 2>&1
timeout 300 python tools/exp/r06_dbg_fallback.py > $O/r06m_dbg_fallback.txt 2>&1
timeout 600 python bench.py > $O/r06m_bench.json 2> $O/r06m_bench.err
timeout 2400 python -m pytest tests -q -m gpu -x > $O/r06m_gpu_suite.txt 2>&1
