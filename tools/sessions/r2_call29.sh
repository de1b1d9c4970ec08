#!/bin/bash
# round 2, GPU call 29: overlapping launches per Analyse instance; latency of small search launches
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_vs_shim.py -x -q -m gpu 2>&1 | tail -3 | tee $out/c29_tests.txt
for b in 6 22 86; do
  timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 --batch $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b (', 6*$b, 'chains):', round(d['roofline']['avg_launch_ms'],1), 'ms per search launch,', round(d['value'],1), 'fps')"
done 2>&1 | tee $out/c29_small_launches.txt
timeout 800 python tools/vs_4k_run.py 36 64 2>&1 | grep -v amdgpu.ids | tee $out/c29_vs_4k_36.txt
timeout 1200 python tools/vs_4k_run.py 144 128 2>&1 | grep -v amdgpu.ids | tee $out/c29_vs_4k_144.txt
