#!/bin/bash
# Contention probe (round 4): the NS step with an EXTRA J-only board kernel on a side stream, forked at
# 1 = after the board kernel, 2 = after the assembly, 3 = after the reduce; joined before the next board kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # label, env...
    local label=$1; shift
    local v=$(env "$@" python $R/bench.py --no-cpu-baseline --no-full-solve --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$label  ms_per_step $v"
}
{
run "baseline           " X=1
run "baseline           " X=1
for f in 1 2 3; do
  run "fork $f             " MRCAL_AMD_EXP_SIDEJ=$f
  run "fork $f lowprio     " MRCAL_AMD_EXP_SIDEJ=$f MRCAL_AMD_EXP_LOWPRIO=1
  run "fork $f cuoff 32 p0 " MRCAL_AMD_EXP_SIDEJ=$f MRCAL_AMD_EXP_CUOFF=32 MRCAL_AMD_EXP_CUPAT=0
  run "fork $f cuoff 32 p1 " MRCAL_AMD_EXP_SIDEJ=$f MRCAL_AMD_EXP_CUOFF=32 MRCAL_AMD_EXP_CUPAT=1
  run "fork $f cuoff 64 p1 " MRCAL_AMD_EXP_SIDEJ=$f MRCAL_AMD_EXP_CUOFF=64 MRCAL_AMD_EXP_CUPAT=1
done
} > $O/exp_sidej.txt 2>&1
for f in 1 3; do
rm -rf /tmp/prof_e$f
MRCAL_AMD_EXP_SIDEJ=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e$f -- python $R/bench.py --no-cpu-baseline --no-full-solve > /dev/null 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_e$f "EXP fork $f" > $O/exp_sidej_stats_f$f.txt
done
cat $O/exp_sidej.txt
