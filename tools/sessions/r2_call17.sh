#!/bin/bash
# round 2, GPU call 17: leftover Super kernels launch only the workgroups that have work
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_vs_shim.py -x -q -m gpu -k "super or shim or finest or pelclip" 2>&1 | tail -4 | tee $out/c17_tests_super.txt
timeout 300 python tools/super_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/c17_super_bench.txt
timeout 300 python tools/super_bench.py 1920 1080 8 512 2>&1 | grep -v amdgpu.ids | tee -a $out/c17_super_bench.txt
