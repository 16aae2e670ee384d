# round 6, call 7: the fallback test, the world-1 RCCL worker, the hand-off probe, counters of the splined rows kernel
O=gpurun_out
python -m pytest tests/test_solver_parity.py tests/test_parallel_gpu.py -q -m gpu -x -s -k "explicit_inverse or world1 or nested_dissection or tail_kernel" > $O/r06g_tests.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/handoff tools/exp/handoff_32k.hip 2>/dev/null
timeout 120 /tmp/handoff > $O/r06g_handoff_32k.txt 2>&1
bash tools/collect_r06_pmc.sh r06g "2" > /dev/null 2>&1
