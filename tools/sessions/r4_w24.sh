#!/bin/bash
# round 4: a whole pass of row loads in flight (24) in the 256-register builds
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "analyse_speculative or two_chains or full_size_parity_cfg3" 2>&1 | tail -3
O=gpurun_out/r4_spec_w24.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O; }
EXTRA="--batch 341"; run "batch 341 (2046 chains, 2 per SIMD, 24 row loads in flight)" A=1
EXTRA="--batch 170"; run "batch 170 (1020 chains, 1 per SIMD, 24 in flight)" A=1
EXTRA="--batch 512"; run "batch 512 (3072 chains, 3 per SIMD, 12 in flight)" A=1
EXTRA="--batch 512"; run "batch 512, two-per-SIMD build (two rounds)" MVX_FAST_K=2
cat $O
