#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrcal_amd/csrc -o /tmp/diag16_bench_0 tools/exp/diag16_bench.hip 2>/dev/null && echo "one wave: $(/tmp/diag16_bench_0)"
for f in 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHOL_PAIR_FAST=$f -I mrcal_amd/csrc -o /tmp/diag16_pair_bench_$f tools/exp/diag16_pair_bench.hip 2>/dev/null && timeout 60 /tmp/diag16_pair_bench_$f
done
