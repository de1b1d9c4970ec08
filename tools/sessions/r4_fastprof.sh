#!/bin/bash
# first measurement of round 4: where one chain of the lean kernel spends its cycles, at three chains per SIMD and alone
#   (before:  python tools/build_variant.py fastprof "MVX_FAST_PROF" mvx_analyse_u16.hip)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp MVX_LIB=$PWD/tools/variants/fastprof.so
mkdir -p gpurun_out
O=gpurun_out/r4_lean_kernel_phase_cycles.txt; : > $O
for b in 512 342 171; do timeout 120 python tools/fastprof.py cfg3 $b 2>&1 | grep -v amdgpu.ids >> $O; echo >> $O; done
cat $O
