#!/bin/bash
# round 4: Degrain luma reads odd-sample blocks from the shifted copy of the super plane (dword-aligned loads): tests, then kernel times with / without
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "degrain or pipeline or cfg3 or shim or shell" 2>&1 | tail -4 > gpurun_out/r4_degrain_shadow_tests.txt
cat gpurun_out/r4_degrain_shadow_tests.txt
O=gpurun_out/r4_degrain_shadow.txt; : > $O
for v in 1 0; do
  (cd /tmp && rm -rf /tmp/kt && MVX_DEGRAIN_SHADOW=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $root/bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 > /tmp/kt.log 2>&1)
  f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1)
  echo "== MVX_DEGRAIN_SHADOW=$v" >> $O
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'degrain' in n or 'analyse_' in n: print(n[:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')" >> $O
  grep '^{' /tmp/kt.log | python3 -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print(round(d['value'],1),'fps',round(d['ms_per_step'],1),'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O
done
cat $O
