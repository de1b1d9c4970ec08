#!/bin/bash
# round 2, GPU call 37: Degrain tile order again, now that its loads are batched (XCD-contiguous order was 8 ms slower before)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r default X=1 --
r degrain-xcd MVX_DEGRAIN_XCD=1 --
r cfg5-default X=1 -- --config cfg5
r cfg5-degrain-xcd MVX_DEGRAIN_XCD=1 -- --config cfg5
} 2>&1 | tee $out/c37_variants.txt
