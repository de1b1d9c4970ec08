#!/bin/bash
# round 2, GPU call 15: line-aligned wave spans in the rows kernels
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "super" 2>&1 | tail -4 | tee $out/c15_tests_super.txt
timeout 300 python tools/super_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/c15_super_bench.txt
timeout 300 python tools/super_bench.py 1920 1080 8 512 2>&1 | grep -v amdgpu.ids | tee -a $out/c15_super_bench.txt
