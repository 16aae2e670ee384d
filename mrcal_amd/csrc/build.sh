#!/bin/bash
# Builds mrcal_amd/libmrcal_amd.so for gfx950 (MI355X). hipcc cross-compiles
# without a GPU present.
#
# Round 6: every source is its own translation unit, compiled side by side (as many at a time as there are cores,
# JOBS=n to say otherwise) into csrc/_build/, then linked: 25 s on eight cores where the one command took 64 (the
# solver's kernels were ONE 6800-line unit until then: solver_kernels.hip, now assembly.hip ... factorization_solve.hip).
#
# Measurement builds: any -D..._TS (kernel-internal cycle stamps) or -DMRCAL_AMD_DEV (the board kernel's ablation
# knob) among the arguments builds mrcal_amd/libmrcal_amd_dev.so INSTEAD, with -DMRCAL_AMD_DEV: the shipped library
# carries none of that code. The dev tools load it through MRCAL_AMD_LIB=<path> (tools/README.md)
# -ffp-contract=on (round 5; hipcc's default is fast): a multiply-add is fused where the source writes a*b + c in ONE
# expression - the front end's decision, the same in every kernel a formula is compiled into - and nowhere else. With
# "fast" the back end fuses what it finds after inlining, and the same projection code rounded differently in the board
# kernel's variants (with / without the Gram): J of optimizer_callback() and J inside the
# factorization it returns differed in their last bits. Cost at the metric's size: +0.4 us of the board kernel's 75
# (the two builds alternating on one box: 181.8 against 180.9 us a trial step, inside the run-to-run spread)
set -e
cd "$(dirname "$0")"
OUT=../libmrcal_amd.so
DEV=
OBJ=_build
for a in "$@"; do
    case "$a" in -DMRCAL_AMD_DEV|-D*_TS) OUT=../libmrcal_amd_dev.so; DEV=-DMRCAL_AMD_DEV; OBJ=_build_dev;; esac
done
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $DEV"
SOURCES="kernels.hip splined_kernels.hip project_kernels.hip cholesky_large.hip assembly_splined.hip schur.hip assembly.hip step.hip factorization_solve.hip cholesky_lds.hip uncertainty.hip
         problem.cpp solver.cpp cabi_layout.cpp factorization.cpp unproject.cpp comm.cpp cameramodel_io.cpp"
JOBS=${JOBS:-$(nproc)}
mkdir -p $OBJ
rm -f $OBJ/*.o $OBJ/*.failed
pids=()
for f in $SOURCES; do
    ( $HIPCC $FLAGS "$@" -x hip -c $f -o $OBJ/${f%.*}.o || touch $OBJ/${f%.*}.failed ) &
    pids+=($!)
    while [ "$(jobs -rp | wc -l)" -ge "$JOBS" ]; do sleep 0.2; done
done
wait
if ls $OBJ/*.failed > /dev/null 2>&1; then echo "build failed: $(ls $OBJ/*.failed)"; exit 1; fi
$HIPCC --offload-arch=gfx950 -fPIC -shared -o $OUT $OBJ/*.o -ldl -pthread
echo "built $(readlink -f $OUT)"
