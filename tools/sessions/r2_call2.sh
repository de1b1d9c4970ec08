#!/bin/bash
# round 2, GPU call 2: parity of the lean kernel / shadow layout, then A/B timing
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "analyse or degrain_parity or golden or smoke" 2>&1 | tail -15 | tee $out/c2_tests.txt
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r lean-shadow-k2 X=1 --
r lean-plain-k2 MVX_SHADOW=0 --
r general-plain MVX_SHADOW=0 MVX_GENERAL=1 --
r general-shadowalloc MVX_GENERAL=1 --
r lean-shadow-k3-b512 X=1 -- --batch 512
r lean-shadow-k4-b672 X=1 -- --batch 672
r lean-shadow-k2-b672 MVX_FAST_WPE=2 -- --batch 672
r lean-plain-k3-b512 MVX_SHADOW=0 -- --batch 512
r lean-shadow-k1-b168 X=1 -- --batch 168
r cfg2-lean X=1 -- --config cfg2
r cfg2-general MVX_GENERAL=1 MVX_SHADOW=0 -- --config cfg2
r cfg5-lean X=1 -- --config cfg5
r cfg5-general MVX_GENERAL=1 MVX_SHADOW=0 -- --config cfg5
} 2>&1 | tee $out/c2_variants.txt
