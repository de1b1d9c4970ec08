#!/bin/bash
# round 4, session 6: does the row pitch (in 128-byte lines) change the search?  (L1 set spread of a block's rows and planes)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
MVX_PITCH_LINES=odd timeout 600 python -m pytest tests -x -q -m gpu -k "super_parity or analyse_speculative or degrain_parity or compensate_parity" 2>&1 | tail -4 | tee gpurun_out/r4_pitch_tests.txt
O=gpurun_out/r4_pitch_bench.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O; }
run "pitch 256-byte multiple (62 lines luma, 32 chroma): default" A=1
run "odd number of lines (61 / 31)" MVX_PITCH_LINES=odd
run "65 lines = 1 mod 64 (rows on consecutive sets)" MVX_PITCH_LINES=1
run "67 lines = 3 mod 64" MVX_PITCH_LINES=3
run "64 lines = 0 mod 64 (every row of a plane on the same set)" MVX_PITCH_LINES=0
run "odd, serial lean kernel" MVX_PITCH_LINES=odd MVX_SPEC=0
cat $O
