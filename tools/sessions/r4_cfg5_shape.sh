#!/bin/bash
# round 4: cfg5 (8K16, 32x32 blocks, serial lean kernel at two chains per SIMD): workgroup size and barrier interval
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_cfg5_launch_shape.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --no-others --no-parity --steps 1 --warmup 1 --config cfg5 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step')" >> $O; }
run "default (workgroups of 8, barrier every 32 blocks)"
run "workgroups of 4, barrier every 32" MVX_FAST_CPW=4
run "workgroups of 4, barrier every 128" MVX_FAST_CPW=4 MVX_CPW_SYNC=128
run "workgroups of 8, barrier every 128" MVX_CPW_SYNC=128
run "workgroups of 8, barrier every 8" MVX_CPW_SYNC=8
run "workgroups of 2, barrier every 32" MVX_FAST_CPW=2
cat $O
