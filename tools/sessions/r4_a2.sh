#!/bin/bash
# round 4: A2 reads the pattern's table entries up front: search parity tests, cfg3 / cfg2 / cfg4
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "speculative or full_size_parity" 2>&1 | tail -4 > gpurun_out/r4_a2_tests.txt
cat gpurun_out/r4_a2_tests.txt
O=gpurun_out/r4_a2_preload.txt; : > $O
run() { echo "== $1" >> $O; shift; c=$1; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --config $c 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O; }
for c in cfg3 cfg2 cfg4; do run "$c" $c; done
cat $O
