#!/bin/bash
# round 4: per-kernel times of cfg4 and cfg2 (what the step spends outside the search)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out
O=gpurun_out/r4_cfg4_cfg2_kernel_times.txt; : > $O
for c in cfg4 cfg2; do
  (cd /tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $root/bench.py --no-cpu --no-parity --no-traffic --no-others --steps 2 --warmup 1 --config $c > /tmp/kt.log 2>&1)
  f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1)
  echo "== $c" >> $O
  python3 -c "
import csv
rows=[r for r in csv.DictReader(open('$f')) if 'at::native' not in r['Name'] and 'rocclr' not in r['Name']]
for r in rows[:18]: print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.3f'%(float(r['TotalDurationNs'])/1e6)).rjust(10), 'ms total', ('%.3f'%(float(r['AverageNs'])/1e6)).rjust(9), 'ms avg')" >> $O
  grep '^{' /tmp/kt.log | python3 -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print(round(d['value'],1),'fps',round(d['ms_per_step'],1),'ms/step', d['config'].get('frames_per_step_per_gpu'))" >> $O
done
cat $O
