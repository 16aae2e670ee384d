#!/bin/bash
# Builds mrcal_amd/libmrcal_amd.so for gfx950 (MI355X). hipcc cross-compiles
# without a GPU present.
set -e
cd "$(dirname "$0")"
OUT=../libmrcal_amd.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -o $OUT -ldl \
    kernels.hip solver_kernels.hip problem.cpp cabi_layout.cpp solver.cpp factorization.cpp unproject.cpp comm.cpp cameramodel_io.cpp uncertainty.hip "$@"
echo "built $(readlink -f $OUT)"
