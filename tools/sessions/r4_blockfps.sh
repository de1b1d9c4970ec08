#!/bin/bash
# round 4: BlockFPS row kernel with vectorised table / mask loads: parity tests, cfg4 bench, kernel times
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "blockfps or cfg4 or BlockFPS" 2>&1 | tail -4 > gpurun_out/r4_blockfps_tests.txt
cat gpurun_out/r4_blockfps_tests.txt
O=gpurun_out/r4_blockfps_rows.txt; : > $O
(cd /tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $root/bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --config cfg4 > /tmp/kt.log 2>&1)
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1)
python3 -c "
import csv
rows=[r for r in csv.DictReader(open('$f')) if 'at::native' not in r['Name'] and 'rocclr' not in r['Name']]
for r in rows[:8]: print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), ('%.3f'%(float(r['AverageNs'])/1e6)).rjust(9), 'ms avg')" >> $O
grep '^{' /tmp/kt.log | python3 -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print(round(d['value'],1),'fps',round(d['ms_per_step'],1),'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O
cat $O
