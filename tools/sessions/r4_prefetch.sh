#!/bin/bash
# round 4: leading-edge prefetch in the row passes
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "analyse_speculative or two_chains or many_chains or full_size_parity_cfg3" 2>&1 | tail -3
O=gpurun_out/r4_spec_prefetch.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O; }
EXTRA="--batch 341"; run "batch 341 (2 per SIMD), prefetch" A=1
EXTRA="--batch 341"; run "batch 341 (2 per SIMD), no prefetch (MVX_SPEC=4)" MVX_SPEC=4
EXTRA="--batch 512"; run "batch 512 (3 per SIMD), prefetch" A=1
cat $O
MVX_LIB=$PWD/tools/variants/specprof.so timeout 200 python tools/specprof.py cfg3 341 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_spec_phase_cycles_prefetch.txt
