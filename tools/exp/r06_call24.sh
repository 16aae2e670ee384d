# round 6, call 24: does the prologue launch's length depend on the step before it? and the line once more, with the traffic files as committed
O=gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ns
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ns -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/exp/prologue_durations.py /tmp/prof_ns > $GRAFT_REPO_ROOT/$O/r06x_prologue_durations.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > $O/r06x_bench.json 2> $O/r06x_bench.err
