#!/bin/bash
# round 4: barrier interval of the speculative kernel's workgroups (groups of 32 blocks), and the batch size
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_spec_sync_interval.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --no-parity --steps 2 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step')" >> $O; }
for s in 32 64 128 256; do run "batch 512 (3072 chains, 3 per SIMD), barrier every $s blocks" MVX_CPW_SYNC=$s; done
EXTRA="--batch 341"
for s in 32 64 128; do run "batch 341 (2046 chains, 2 per SIMD), barrier every $s blocks" MVX_CPW_SYNC=$s; done
cat $O
