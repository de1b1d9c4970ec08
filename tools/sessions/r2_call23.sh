#!/bin/bash
# round 2, GPU call 23: filter shell with per-thread streams and pinned staging copies
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_vs_shim.py -x -q -m gpu 2>&1 | tail -3 | tee $out/c23_tests.txt
timeout 1200 python tools/vs_4k_run.py 144 128 2>&1 | grep -v amdgpu.ids | tee $out/c23_vs_4k_144.txt
