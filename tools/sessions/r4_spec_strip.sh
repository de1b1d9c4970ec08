#!/bin/bash
# round 4, session 7: pass A with runs (strips of seven blocks that share a displacement) -- parity, bench with and without runs
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "analyse or golden or smoke or degrain_parity or full_size_parity_cfg3 or full_size_parity_cfg5" 2>&1 | tail -8 | tee gpurun_out/r4_spec_strip_tests.txt
O=gpurun_out/r4_spec_strip_bench.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O; }
run "runs of seven blocks (default)" A=1
run "no runs (MVX_SPEC=3)" MVX_SPEC=3
run "runs, two chains per SIMD" MVX_FAST_K=2
run "serial lean kernel" MVX_SPEC=0
cat $O
