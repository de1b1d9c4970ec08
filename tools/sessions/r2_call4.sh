#!/bin/bash
# round 2, GPU call 4: locality experiments for the lean kernel (sync interval, XCD-contiguous workgroups, one reference frame per workgroup)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "analyse or degrain_parity or golden" 2>&1 | tail -4 | tee $out/c4_tests.txt
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r k2-base X=1 --
r k2-nodegrainshadow MVX_DEGRAIN_SHADOW=0 --
r k2-sync16 MVX_CPW_SYNC=16 --
r k2-sync64 MVX_CPW_SYNC=64 --
r k2-xcd MVX_FAST_FLAGS=1 --
r k2-pad MVX_PAD_RUNS=1 --
r k2-pad-sync16 MVX_PAD_RUNS=1 MVX_CPW_SYNC=16 --
r k2-lumashadow MVX_SHADOW_PLANES=1 --
r k3-base X=1 -- --batch 512
r k3-sync16 MVX_CPW_SYNC=16 -- --batch 512
r k3-sync64 MVX_CPW_SYNC=64 -- --batch 512
r k3-xcd MVX_FAST_FLAGS=1 -- --batch 512
r k3-cpw6-sync16 MVX_FAST_CPW=6 MVX_CPW_SYNC=16 -- --batch 512
r k3-cpw6 MVX_FAST_CPW=6 -- --batch 512
} 2>&1 | tee $out/c4_variants.txt
