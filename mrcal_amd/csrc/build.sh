#!/bin/bash
# Builds mrcal_amd/libmrcal_amd.so for gfx950 (MI355X). hipcc cross-compiles
# without a GPU present.
#
# Measurement builds: any -D..._TS (kernel-internal cycle stamps) or -DMRCAL_AMD_DEV (the board kernel's ablation
# knob) among the arguments builds mrcal_amd/libmrcal_amd_dev.so INSTEAD, with -DMRCAL_AMD_DEV: the shipped library
# carries none of that code. The dev tools load it through MRCAL_AMD_LIB=<path> (tools/README.md)
# -ffp-contract=on (round 5; hipcc's default is fast): a multiply-add is fused where the source writes a*b + c in ONE
# expression - the front end's decision, the same in every kernel a formula is compiled into - and nowhere else. With
# "fast" the back end fuses what it finds after inlining, and the same projection code rounded differently in the board
# kernel's variants (with / without the Gram; the one-launch form): J of optimizer_callback() and J inside the
# factorization it returns differed in their last bits. Cost at the metric's size: +0.4 us of the board kernel's 75
# (the two builds alternating on one box: 181.8 against 180.9 us a trial step, inside the run-to-run spread)
set -e
cd "$(dirname "$0")"
OUT=../libmrcal_amd.so
DEV=
for a in "$@"; do
    case "$a" in -DMRCAL_AMD_DEV|-D*_TS) OUT=../libmrcal_amd_dev.so; DEV=-DMRCAL_AMD_DEV;; esac
done
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on \
    -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -o $OUT -ldl -pthread $DEV \
    kernels.hip solver_kernels.hip problem.cpp cabi_layout.cpp solver.cpp factorization.cpp unproject.cpp comm.cpp cameramodel_io.cpp uncertainty.hip "$@"
echo "built $(readlink -f $OUT)"
