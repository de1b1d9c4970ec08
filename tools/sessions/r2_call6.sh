#!/bin/bash
# round 2, GPU call 6: whole GPU suite (new block sizes, Degrain4/5, concurrent shell requests), default benches
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $out/c6_tests.txt
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r cfg3-default X=1 --
r cfg2-default X=1 -- --config cfg2
r cfg5-default X=1 -- --config cfg5
r cfg1-default X=1 -- --config cfg1
} 2>&1 | tee $out/c6_variants.txt
