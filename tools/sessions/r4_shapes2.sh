#!/bin/bash
# round 4: launch shapes of the two-stage kernel: three chains per SIMD, two rounds, four-chain workgroups
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_spec_twostage_shapes.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'), d['config']['chains_per_step_per_gpu'], 'chains')" >> $O; }
EXTRA=""; run "batch 341, two per SIMD, workgroups of 8 (default)" A=1
EXTRA=""; run "batch 341, workgroups of 4" MVX_FAST_CPW=4
EXTRA="--batch 682"; run "batch 682 (two rounds)" A=1
EXTRA="--batch 512"; run "batch 512, three per SIMD" MVX_FAST_K=3
cat $O
