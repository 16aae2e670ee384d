#!/bin/bash
# Builds mrcal_amd/libmrcal_amd.so for gfx950 (MI355X). hipcc cross-compiles
# without a GPU present.
#
# Measurement builds: any -D..._TS (kernel-internal cycle stamps) or -DMRCAL_AMD_DEV (the board kernel's ablation
# knob) among the arguments builds mrcal_amd/libmrcal_amd_dev.so INSTEAD, with -DMRCAL_AMD_DEV: the shipped library
# carries none of that code. The dev tools load it through MRCAL_AMD_LIB=<path> (tools/README.md)
set -e
cd "$(dirname "$0")"
OUT=../libmrcal_amd.so
DEV=
for a in "$@"; do
    case "$a" in -DMRCAL_AMD_DEV|-D*_TS) OUT=../libmrcal_amd_dev.so; DEV=-DMRCAL_AMD_DEV;; esac
done
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -o $OUT -ldl $DEV \
    kernels.hip solver_kernels.hip problem.cpp cabi_layout.cpp solver.cpp factorization.cpp unproject.cpp comm.cpp cameramodel_io.cpp uncertainty.hip "$@"
echo "built $(readlink -f $OUT)"
