#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_r05.sh [tag] [parts]'): everything profiles/r05* is made of.
# parts: any of  bench stats configs pmc traffic mfma ns   (default: all but ns = the 3-minute CPU reference solve)
tag=${1:-r05}
parts=${2:-"bench stats configs pmc traffic mfma"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
has() { [[ " $parts " == *" $1 "* ]]; }

if has bench; then
    python $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
fi
if has stats; then
    rm -rf /tmp/prof_ns /tmp/prof_c2 /tmp/prof_c3
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns -- python $R/bench.py --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
    python $R/tools/kernel_stats_table.py /tmp/prof_ns "round 5 ($tag), 8 cameras x 1000 frames OPENCV8: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-full-solve --no-configs (30 scratch + 5 warmup + 50 timed steps)" > $O/${tag}_kernel_stats.txt
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/tools/probe_config2.py > $O/${tag}_config2.log 2>&1
    python $R/tools/kernel_stats_table.py /tmp/prof_c2 "round 5 ($tag), configuration 2 (1 camera x 800 frames, SPLINED 30x20 over 150 degrees, core locked; seed 4): rocprofv3 --kernel-trace --stats -- python tools/probe_config2.py (12 trial steps + one full solve)" > $O/${tag}_kernel_stats_config2_splined.txt
    python $R/tools/step_trace_dump.py /tmp/prof_c2 8 > $O/${tag}_config2_step_in_time_order.txt 2>&1
    python $R/tools/exp/lchol_launches.py /tmp/prof_c2 8 > $O/${tag}_config2_lchol_launches.txt 2>&1
    python $R/tools/step_trace_dump.py /tmp/prof_ns 41 > $O/${tag}_ns_step_in_time_order.txt 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/tools/probe_configs.py 3 > $O/${tag}_config3.log 2>&1
    python $R/tools/kernel_stats_table.py /tmp/prof_c3 "round 5 ($tag), configuration 3 (16 cameras x 2000 frames OPENCV8, one GPU): rocprofv3 --kernel-trace --stats -- python tools/probe_configs.py 3 (23 trial steps + one full solve)" > $O/${tag}_kernel_stats_config3.txt
fi
if has configs; then
    python $R/tools/probe_configs.py > $O/${tag}_configs_table.md 2> $O/${tag}_configs.err
fi
if has pmc; then
    # the board kernel's instruction / cycle counters, 4 per pass (each pass with --kernel-trace only)
    rm -rf /tmp/pmc_b*
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
               "SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
               "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
               "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
               "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
        timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_b$i -- python $R/tools/probe_board_one.py 1 0 3 > /dev/null 2>&1
        i=$((i+1))
    done
    mkdir -p /tmp/pmc_ball; cp -r /tmp/pmc_b[0-9]* /tmp/pmc_ball/ 2>/dev/null
    python $R/tools/pmc_summary.py /tmp/pmc_ball "board_kernel<0, 8, true, true, true>" > $O/${tag}_board_kernel_pmc_raw.txt 2>&1
fi
if has traffic; then
    rm -rf /tmp/pmc_w /tmp/pmc_f
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/tools/probe_board_one.py 1 0 5 > /dev/null 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/tools/probe_board_one.py 1 0 5 > /dev/null 2>&1
    { python $R/tools/pmc_summary.py /tmp/pmc_w "board_kernel<0, 8, true, true, true>"; python $R/tools/pmc_summary.py /tmp/pmc_f "board_kernel<0, 8, true, true, true>"; } > $O/${tag}_board_kernel_traffic_raw.txt 2>&1
fi
if has mfma; then
    PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"
    rm -rf /tmp/pmc_ns /tmp/pmc_c2
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_ns -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-solve --no-configs > /dev/null 2>&1
    python $R/tools/mfma_util.py /tmp/pmc_ns $O/${tag}_mfma_ns.json board_kernel schur_syrk_mfma_kernel schur_cholesky_solve_kernel > /dev/null
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_c2 -- python $R/tools/probe_config2.py > /dev/null 2>&1
    python $R/tools/mfma_util.py /tmp/pmc_c2 $O/${tag}_mfma_config2.json schur_syrk_sparse_kernel lchol_panel_kernel assemble_splined_kernel > /dev/null
fi
if has ns; then
    python $R/tools/ns_solve_vs_reference.py $O/${tag}_ns_solve_vs_reference.json > $O/${tag}_ns_solve.log 2>&1
fi
ls -la $O | grep $tag
