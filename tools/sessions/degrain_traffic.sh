#!/bin/bash
# FETCH_SIZE of the Degrain cell kernels on a clip whose vectors are all ~0 (aligned, every covering block reads the same
# lines) against the bench clip (vectors of +-6 .. +-18 half-pel units).  usage: gpurun -- 'bash tools/gpu_session.sh sh tools/sessions/degrain_traffic.sh'
export TMPDIR=/tmp
out=$PWD/gpurun_out; root=$PWD
: > $out/r3_degrain_traffic.txt
for kind in moving static; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/dg_$c
    (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/dg_$c -o p -- python $root/tools/degrain_traffic.py $kind 64 > /tmp/dg_${kind}_$c.log 2>&1)
  done
  grep "algorithmic" /tmp/dg_${kind}_WRITE_SIZE.log 2>/dev/null | tail -1 >> $out/r3_degrain_traffic.txt
  python3 - $kind >> $out/r3_degrain_traffic.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('/tmp/dg_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c or 'degrain' not in r['Kernel_Name']: continue
            k = r['Kernel_Name'].split('(')[0]
            acc[k][c] += float(r['Counter_Value']); n[(k, c)] += 1
for k in acc:
    fs = acc[k]['FETCH_SIZE'] / max(1, n[(k, 'FETCH_SIZE')]); ws = acc[k]['WRITE_SIZE'] / max(1, n[(k, 'WRITE_SIZE')])
    print("%-8s %-50s FETCH_SIZE %.2f GB raw (x2 = %.2f GB)  WRITE_SIZE %.2f GB per dispatch" % (sys.argv[1], k, fs * 1024 / 1e9, 2 * fs * 1024 / 1e9, ws * 1024 / 1e9))
PY
done
cat $out/r3_degrain_traffic.txt
