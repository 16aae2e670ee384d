# round 6, call 25: the clearing workgroups of the prologue launch without the choice's sums
O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > $O/r06y_gpu_suite.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ns
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ns -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/exp/prologue_durations.py /tmp/prof_ns > $GRAFT_REPO_ROOT/$O/r06y_prologue_durations.txt 2>&1
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-cpu-baseline > $O/r06y_bench_$i.json 2> /dev/null; done
