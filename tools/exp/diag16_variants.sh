#!/bin/bash
# chol_factor_diag16() variants timed alone (tools/exp/diag16_bench.hip): cycles per call
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHOL_DIAG_VARIANT=$v -I mrcal_amd/csrc -o /tmp/diag16_bench_$v tools/exp/diag16_bench.hip 2>/dev/null && echo "variant $v: $(/tmp/diag16_bench_$v)"
done
