#!/bin/bash
# round 2, GPU call 34: throughput against chains per SIMD, final kernel (1, 2, 3 per SIMD at 1020 / 2046 / 3072 chains)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r k1-1020chains MVX_FAST_WPE=1 -- --batch 170
r k2-2046chains MVX_FAST_WPE=2 -- --batch 341
r k3-3072chains MVX_FAST_WPE=3 -- --batch 512
r k2-1020chains MVX_FAST_WPE=2 -- --batch 170
r k3-1020chains MVX_FAST_WPE=3 -- --batch 170
} 2>&1 | tee $out/c34_variants.txt
