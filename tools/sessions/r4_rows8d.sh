#!/bin/bash
# round 4: 8-bit row passes with dword-aligned loads (rows shifted into place in registers): tests, then cfg2 / cfg4 / cfg1
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "speculative_kernel_8bit" 2>&1 | tail -5 > gpurun_out/r4_rows8_tests.txt
cat gpurun_out/r4_rows8_tests.txt
O=gpurun_out/r4_rows8_aligned.txt; : > $O
run() { echo "== $1" >> $O; shift; c=$1; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --config $c 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'), d['config'].get('chains_per_step_per_gpu'))" >> $O; }
for c in cfg2 cfg4 cfg1; do run "$c row passes, dword-aligned loads" $c; done
cat $O
