#!/bin/bash
# round 2, GPU call 33: four chains per SIMD for 16-bit 16x16 again (21 spilled VGPRs after the load-stream change; 44 before)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r k3-b683 MVX_FAST_WPE=3 -- --batch 683
r k4-b683 MVX_FAST_WPE=4 -- --batch 683
r k4-b683-sync64 MVX_FAST_WPE=4 MVX_CPW_SYNC=64 -- --batch 683
} 2>&1 | tee $out/c33_variants.txt
