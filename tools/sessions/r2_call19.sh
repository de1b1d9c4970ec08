#!/bin/bash
# round 2, GPU call 19: filter shell with the adaptive combining wait
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_vs_shim.py -x -q -m gpu 2>&1 | tail -3 | tee $out/c19_tests.txt
timeout 900 python tools/vs_4k_run.py 36 64 2>&1 | grep -v amdgpu.ids | tee $out/c19_vs_4k.txt
