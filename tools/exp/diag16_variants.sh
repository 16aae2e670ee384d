#!/bin/bash
# (round 4: the variant code this script compiled - CHOL_DIAG_VARIANT 0..3 in chol_factor_diag16() - lived in commit 14901cf and
#  was taken out of the product source again in 88ecca6; the numbers are in profiles/r04_diag16_variants.txt. As it stands the script
#  times the shipped block four times.)
# chol_factor_diag16() variants timed alone (tools/exp/diag16_bench.hip): cycles per call
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHOL_DIAG_VARIANT=$v -I mrcal_amd/csrc -o /tmp/diag16_bench_$v tools/exp/diag16_bench.hip 2>/dev/null && echo "variant $v: $(/tmp/diag16_bench_$v)"
done
