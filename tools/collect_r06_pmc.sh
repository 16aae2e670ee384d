#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_r06_pmc.sh [tag] [configs]'): the Jacobian kernels' counters at the
# configurations that under-fill the chip: instruction / wait counters in passes of 4, then HBM traffic
# (WRITE_SIZE, FETCH_SIZE: a pass each, --kernel-trace only beside them). configs: any of ns 1 2 3 5
tag=${1:-r06}
cfgs=${2:-"1 2 5"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in $cfgs; do
    case $c in 2) K="board_splined";; 5) K="board_tri_kernel";; *) K="board_kernel";; esac
    rm -rf /tmp/pmc_${c}_*
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
               "SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
               "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
               "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
               "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
               "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
               "WRITE_SIZE" "FETCH_SIZE"; do
        timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_${c}_$i -- python $R/tools/probe_board_one.py 1 0 5 $c > /tmp/pmc_${c}_$i.log 2>&1
        i=$((i+1))
    done
    mkdir -p /tmp/pmc_${c}_all; cp -r /tmp/pmc_${c}_[0-9]* /tmp/pmc_${c}_all/ 2>/dev/null
    { echo "# config $c: rocprofv3 --kernel-trace --pmc <set> -- python tools/probe_board_one.py 1 0 5 $c  (a pass per set of 4; WRITE_SIZE and FETCH_SIZE a pass each)"
      tail -1 /tmp/pmc_${c}_0.log
      python $R/tools/pmc_summary.py /tmp/pmc_${c}_all "$K"; } > $O/${tag}_config${c}_jacobian_kernel_pmc_raw.txt 2>&1
done
ls -la $O | grep $tag
