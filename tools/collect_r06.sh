#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_r06.sh [tag] [parts]'): everything profiles/r06* is made of.
# parts: any of  bench stats pmc traffic mfma probes records     (default: all)
tag=${1:-r06}
parts=${2:-"bench stats pmc traffic mfma probes records"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
has() { [[ " $parts " == *" $1 "* ]]; }

if has bench; then
    python $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
fi
if has stats; then
    rm -rf /tmp/prof_ns
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns -- python $R/bench.py --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
    python $R/tools/kernel_stats_table.py /tmp/prof_ns "round 6 ($tag), 8 cameras x 1000 frames OPENCV8: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-full-solve --no-configs (30 scratch + 5 warmup + 50 timed + 16 event-timed steps)" > $O/${tag}_kernel_stats.txt
    python $R/tools/step_trace_dump.py /tmp/prof_ns 41 > $O/${tag}_ns_step_in_time_order.txt 2>&1
    for c in 1 2 3 5; do
        rm -rf /tmp/prof_c$c
        rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c$c -- python $R/bench.py --only-config $c > /dev/null 2>&1
        python $R/tools/kernel_stats_table.py /tmp/prof_c$c "round 6 ($tag), BASELINE.json configuration $c as bench.py's configs[] makes it: rocprofv3 --kernel-trace --stats -- python bench.py --only-config $c (3 + 20 + 12 trial steps, four full solves, 20 steps without the Jacobian stream)" > $O/${tag}_kernel_stats_config$c.txt
    done
    python $R/tools/step_trace_dump.py /tmp/prof_c2 8 > $O/${tag}_config2_step_in_time_order.txt 2>&1
    python $R/tools/exp/lchol_launches.py /tmp/prof_c2 8 > $O/${tag}_config2_lchol_launches.txt 2>&1
fi
if has pmc; then
    # the board kernel's counters at the metric's size (4 per pass), and the Jacobian kernels' where the chip is under-filled
    bash $R/tools/collect_r06_pmc.sh ${tag} "ns 1 2 5" > /dev/null 2>&1
fi
if has mfma; then
    PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"
    rm -rf /tmp/pmc_ns /tmp/pmc_c2
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_ns -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
    python $R/tools/mfma_util.py /tmp/pmc_ns $O/${tag}_mfma_ns.json board_kernel schur_syrk_mfma_kernel schur_cholesky_solve_kernel > /dev/null
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_c2 -- python $R/bench.py --only-config 2 > /dev/null 2>&1
    python $R/tools/mfma_util.py /tmp/pmc_c2 $O/${tag}_mfma_config2.json schur_syrk_sparse_kernel lchol_panel_kernel lchol_nd_pair_kernel assemble_splined_kernel > /dev/null
fi
if has probes; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_rate $R/tools/exp/dispatch_rate.hip 2>/dev/null && timeout 120 /tmp/dispatch_rate > $O/${tag}_dispatch_rate.txt 2>&1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/handoff $R/tools/exp/handoff_32k.hip 2>/dev/null && timeout 120 /tmp/handoff > $O/${tag}_handoff_32k.txt 2>&1
    if [ -f $R/mrcal_amd/libmrcal_amd_dev.so ]; then
        (cd $R; MRCAL_AMD_LIB=mrcal_amd/libmrcal_amd_dev.so python tools/probe_board_ts.py 1 > $O/${tag}_board_ts_config1.txt 2>&1)
    fi
fi
if has records; then
    (cd $R; python tools/solve_vs_recorded_reference.py $O config3 config5 > $O/${tag}_solve_vs_reference.log 2>&1
     python -m pytest tests/test_parallel_gpu.py -q -m gpu -s -k "references_record or world1" > $O/${tag}_records_tests.txt 2>&1)
fi
ls -la $O | grep $tag
