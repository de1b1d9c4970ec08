#!/bin/bash
# round 2, GPU call 8: shell / compensate / blockfps tests with the row kernels, cfg4 bench, 4K16 shell run, r2 profile round (cfg3)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_vs_shim.py tests/test_gpu_parity.py -x -q -m gpu -k "concurrent or cfg4 or shim or compensate or blockfps" 2>&1 | tail -6 | tee $out/c8_tests.txt
timeout 600 python bench.py --config cfg4 --steps 2 --warmup 1 2>&1 | tail -1 | tee $out/c8_cfg4.json | cut -c1-300
timeout 900 python tools/vs_4k_run.py 36 64 2>&1 | tail -8 | tee $out/c8_vs4k.txt
bash tools/profile_round.sh r2 > $out/c8_profile_round.log 2>&1; tail -12 $out/c8_profile_round.log | cut -c1-600
