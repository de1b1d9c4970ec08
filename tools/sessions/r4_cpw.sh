#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_spec_twostage_cpw_sync.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O; }
EXTRA=""
for y in 64 128 256 512; do run "workgroups of 4, barrier every $y blocks" MVX_FAST_CPW=4 MVX_CPW_SYNC=$y; done
run "workgroups of 8, barrier every 128 blocks" MVX_CPW_SYNC=128
cat $O
