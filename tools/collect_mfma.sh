#!/bin/bash
# On the GPU box: matrix-pipe busy share of the MFMA kernels (counters in their own pass: no other trace domain)
# -> gpurun_out/mfma_ns.json, gpurun_out/mfma_config2.json; tools/merge_mfma.py puts them into profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_ns /tmp/pmc_c2
timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_ns -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-solve > /dev/null 2>&1
python $R/tools/mfma_util.py /tmp/pmc_ns $R/gpurun_out/mfma_ns.json board_kernel schur_syrk_mfma_kernel schur_cholesky_solve_kernel > /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_c2 -- python $R/tools/probe_config2.py > /dev/null 2>&1
python $R/tools/mfma_util.py /tmp/pmc_c2 $R/gpurun_out/mfma_config2.json schur_syrk_sparse_kernel lchol_panel_kernel lchol_diag_kernel > /dev/null
ls -la $R/gpurun_out/mfma_*.json
