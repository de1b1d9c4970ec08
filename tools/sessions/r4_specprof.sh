#!/bin/bash
# round 4: where one chain of the speculative kernel spends its cycles (three and two chains per SIMD)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp MVX_LIB=$PWD/tools/variants/specprof.so
mkdir -p gpurun_out
O=gpurun_out/r4_spec_phase_cycles.txt; : > $O
for b in 341; do timeout 200 python tools/specprof.py cfg3 $b 2>&1 | grep -v amdgpu.ids >> $O; echo >> $O; done
cat $O
