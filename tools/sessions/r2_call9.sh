#!/bin/bash
# round 2, GPU call 9: UV-interleaved chroma shadow plane + Degrain tile order
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $out/c9_tests.txt
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r cfg3-uv X=1 --
r cfg3-uv-degrain-noxcd MVX_DEGRAIN_XCD=0 --
r cfg3-lumashadow-only MVX_SHADOW_PLANES=1 --
r cfg3-uv-sync16 MVX_CPW_SYNC=16 --
r cfg3-uv-k2 X=1 -- --batch 336
r cfg5-uv X=1 -- --config cfg5
} 2>&1 | tee $out/c9_variants.txt
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o p -- python $OLDPWD/bench.py --no-cpu --steps 1 --warmup 0 > /tmp/pmc_f.log 2>&1)
python3 - <<'PY' | tee gpurun_out/c9_fetch.txt
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('/tmp/pmc_f/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            k = r['Kernel_Name'].split('(')[0][:60]; acc[k] += float(r['Counter_Value']); n[k] += 1
for k in acc:
    if any(t in k for t in ('analyse', 'degrain', 'super')): print(k, 'FETCH_SIZE GB per dispatch (raw)', round(acc[k] / n[k] / 1e6, 1))
PY
