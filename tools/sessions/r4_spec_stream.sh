#!/bin/bash
# round 4, session 4: pass A as streams of loads that stay in flight across blocks -- parity, bench at 3 / 2 / 1 chains per SIMD, no-live-blocks ablation
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "analyse or golden or smoke or degrain_parity or full_size_parity" 2>&1 | tail -8 | tee gpurun_out/r4_spec_stream_tests.txt
O=gpurun_out/r4_spec_stream_bench.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', 'parity', d.get('parity_check',{}).get('identical'))" >> $O; }
run "default (3 per SIMD)" A=1
run "two chains per SIMD (MVX_FAST_K=2)" MVX_FAST_K=2
run "one chain per SIMD (MVX_FAST_K=1)" MVX_FAST_K=1
run "abl3: every speculative result taken (no live blocks; results wrong)" MVX_LIB=$PWD/tools/variants/abl3.so
run "serial lean kernel (MVX_SPEC=0)" MVX_SPEC=0
cat $O
