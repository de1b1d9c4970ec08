#!/bin/bash
# round 2, GPU call 11: whole suite, benches with the batched Degrain loads and the search micro-optimisations
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $out/c11_tests.txt
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r cfg3 X=1 --
r cfg5 X=1 -- --config cfg5
r cfg2 X=1 -- --config cfg2
} 2>&1 | tee $out/c11_variants.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $OLDPWD/bench.py --no-cpu --steps 2 --warmup 1 > /tmp/kt.log 2>&1)
python3 - <<'PY'
import glob
f = glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True)
if f:
    rows = open(f[0]).read().splitlines()
    open('gpurun_out/c11_kernel_stats.csv', 'w').write("\n".join(rows))
    for r in rows[:9]: print(r[:150])
PY
