#!/bin/bash
# round 4: Degrain cell kernel, covering blocks pipelined (MVX_DG_PIPE = 2 / 4 blocks per round) against the default build: kernel times from rocprofv3
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out
O=gpurun_out/r4_degrain_pipe.txt; : > $O
for v in default dgpipe2 dgpipe4; do
  lib=""; [ $v != default ] && lib=$root/tools/variants/$v.so
  (cd /tmp && rm -rf /tmp/kt && MVX_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $root/bench.py --no-cpu --no-parity --no-traffic --no-others --steps 2 --warmup 1 > /tmp/kt.log 2>&1)
  f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1)
  echo "== $v" >> $O
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'degrain' in n or 'analyse_' in n or 'super' in n: print(n[:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms avg')" >> $O
  grep '^{' /tmp/kt.log | python3 -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print(round(d['value'],1),'fps',round(d['ms_per_step'],1),'ms/step')" >> $O
done
cat $O
