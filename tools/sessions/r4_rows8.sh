#!/bin/bash
# round 4: the 8-bit row passes (8x8 blocks overlapping by four): parity tests, then cfg2 against the serial lean kernel and at 2 / 3 chains per SIMD
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "speculative_kernel_8bit" 2>&1 | tail -15 > gpurun_out/r4_rows8_tests.txt
cat gpurun_out/r4_rows8_tests.txt
O=gpurun_out/r4_rows8_cfg2.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --config cfg2 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'), d['config'].get('chains_per_step_per_gpu'))" >> $O; }
run "serial lean kernel" MVX_SPEC=0
run "row passes, default launch shape" MVX_SPEC=1
run "row passes, 3 chains per SIMD" MVX_SPEC=1 MVX_FAST_K=3
run "row passes, workgroups of 8" MVX_SPEC=1 MVX_FAST_CPW=8
run "spec kernel without row passes" MVX_SPEC=3
cat $O
