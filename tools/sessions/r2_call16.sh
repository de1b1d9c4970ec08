#!/bin/bash
# round 2, GPU call 16: per-kernel times of mv.Super alone (1080p 8 bit, 4K 16 bit)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
for cfg in "1920 1080 8 512" "3840 2160 16 96"; do
  tag=$(echo $cfg | tr ' ' '_')
  (cd /tmp && rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $OLDPWD/tools/super_bench.py $cfg > /tmp/kt.log 2>&1)
  python3 - $tag <<'PY'
import glob, sys
f = glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True)
if f:
    rows = open(f[0]).read().splitlines()
    open('gpurun_out/c16_super_kernel_stats_%s.csv' % sys.argv[1], 'w').write("\n".join(rows))
    for r in rows[:16]: print(r[:160])
PY
done
