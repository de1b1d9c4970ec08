#!/bin/bash
# usage: tools/pmc.sh "<counter group 1>" "<counter group 2>" ... -- <command>
# One rocprofv3 --pmc pass per group (kernel-trace only), summarised per kernel into gpurun_out/pmc_summary.txt
cd /tmp && export TMPDIR=/tmp
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done; shift
out=/root/repo/gpurun_out; mkdir -p $out; : > $out/pmc_summary.txt
i=0
for g in "${groups[@]}"; do
  d=/tmp/pmc_$i; rm -rf $d
  (cd /root/repo && timeout 600 rocprofv3 --pmc $g --kernel-trace --kernel-include-regex "${PMC_KERNELS:-analyse_|degrain|super_|compensate|blockfps|bf_|usable}" --output-format csv -d $d -o p -- "$@" > /tmp/pmc_$i.log 2>&1)
  python3 - $d "$g" >> $out/pmc_summary.txt <<'PY'
import sys, csv, glob, collections
d, g = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
print('## group:', g)
for k in acc:
    import os, re
    flt = os.environ.get('PMC_FILTER')  # regex over kernel names (default: the search, the per-sample Degrain gather, the level-0 Super kernel)
    if flt:
        if not re.search(flt, k): continue
    elif 'analyse' not in k and 'degrain_kernel' not in k and 'super_level0' not in k: continue
    for c, v in acc[k].items():
        print(f'{k:60s} {c:36s} total {v:.6g} dispatches {n[(k,c)]} per-dispatch {v/n[(k,c)]:.6g}')
PY
  i=$((i+1))
done
cat $out/pmc_summary.txt; tail -3 /tmp/pmc_0.log
