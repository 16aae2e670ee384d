#!/bin/bash
# round 5, first GPU call: the driver's line with configs[], configuration 2's kernel table and a step in time order,
# the two new switches A/B on one box, the tests the round's first changes touch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time python $R/bench.py > $O/r05a_bench.json 2> $O/r05a_bench.err ) 2> $O/r05a_bench.time
tail -c 600 $O/r05a_bench.err
rm -rf /tmp/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/tools/probe_config2.py > $O/r05a_c2.log 2>&1
python $R/tools/kernel_stats_table.py /tmp/prof_c2 "round 5 (r05a), configuration 2 (1 camera x 800 frames, SPLINED 30x20 over 150 degrees, core locked): rocprofv3 --kernel-trace --stats -- python tools/probe_config2.py" > $O/r05a_kernel_stats_config2_splined.txt
python $R/tools/step_trace_dump.py /tmp/prof_c2 8 > $O/r05a_config2_step_in_time_order.txt 2>&1
cat $O/r05a_config2_step_in_time_order.txt | cut -c1-100
cd $R
for i in 1 2; do
  for v in "" "MRCAL_AMD_LCHOL_SEPARATE_FINISH=1" "MRCAL_AMD_SPL_PAIRS_LATE=1" "MRCAL_AMD_LCHOL_SEPARATE_FINISH=1 MRCAL_AMD_SPL_PAIRS_LATE=1"; do
    echo "[$v] $(env $v python tools/probe_config2.py 2>&1 | grep 'config2 ms' | cut -c1-44)"
  done
done | tee $O/r05a_config2_switches.txt
timeout 1500 python -m pytest tests/test_full_size.py tests/test_solver_parity.py tests/test_factorization_project.py -x -q -m gpu -k "splined or Jt_x or bare or reproducible" 2>&1 | tail -4 | tee $O/r05a_tests.txt
