#!/bin/bash
# round 4: every BASELINE configuration through the speculative kernel (default) and through the serial lean kernel (MVX_SPEC=0)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_configs_spec_vs_serial.txt; : > $O
for c in cfg3 cfg2 cfg4 cfg5 cfg1; do for s in 1 0; do
  echo "== $c MVX_SPEC=$s" >> $O
  MVX_SPEC=$s timeout 500 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 --config $c 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'), d['config'].get('chains_per_step_per_gpu'))" >> $O
done; done
cat $O
