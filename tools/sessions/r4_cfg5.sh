#!/bin/bash
# round 4: row passes for 32x32 blocks (cfg5)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "analyse or golden or full_size_parity_cfg5 or full_size_parity_cfg3" 2>&1 | tail -5 | tee gpurun_out/r4_cfg5_tests.txt
O=gpurun_out/r4_cfg5_rowpasses.txt; : > $O
for s in 1 0; do
  echo "== cfg5 MVX_SPEC=$s" >> $O
  MVX_SPEC=$s timeout 500 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 --config cfg5 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'), d['roofline']['kernel'][:40])" >> $O
done
cat $O
