#!/bin/bash
# round 4: where one chain of the speculative kernel spends its cycles on 8-bit clips (cfg2, cfg4)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp MVX_LIB=$PWD/tools/variants/specprof8.so
mkdir -p gpurun_out
O=gpurun_out/r4_spec_phase_cycles_8bit.txt; : > $O
for c in cfg2; do echo "== $c" >> $O; timeout 300 python tools/specprof.py $c 2>&1 | grep -v amdgpu.ids >> $O; echo >> $O; done
cat $O
