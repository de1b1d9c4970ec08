#!/bin/bash
# round 2, GPU call 5: barrier-interval sweep at three chains per SIMD, other configurations
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r k3-sync4 MVX_CPW_SYNC=4 -- --batch 512
r k3-sync8 MVX_CPW_SYNC=8 -- --batch 512
r k3-sync32 MVX_CPW_SYNC=32 -- --batch 512
r k3-sync128 MVX_CPW_SYNC=128 -- --batch 512
r k3-sync16-xcd MVX_CPW_SYNC=16 MVX_FAST_FLAGS=1 -- --batch 512
r k3-sync16-b1024 MVX_CPW_SYNC=16 -- --batch 1024 --steps 1
r cfg5-k2-sync16 MVX_CPW_SYNC=16 -- --config cfg5
r cfg5-k2-sync64 MVX_CPW_SYNC=64 -- --config cfg5
r cfg2-k3-sync16 MVX_CPW_SYNC=16 -- --config cfg2
r cfg2-k3-sync64 MVX_CPW_SYNC=64 -- --config cfg2
r cfg2-k4-b2048 X=1 -- --config cfg2 --batch 2048
r cfg2-k4-b2048-sync16 MVX_CPW_SYNC=16 -- --config cfg2 --batch 2048
} 2>&1 | tee $out/c5_variants.txt
