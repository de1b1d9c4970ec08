#!/bin/bash
# round 4, session 2: the speculative search kernel's first run -- parity of everything that searches, verification rates, a first bench line
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "analyse or golden or smoke or degrain_parity or full_size_parity_cfg3" 2>&1 | tail -15 | tee gpurun_out/r4_spec_tests.txt
MVX_LIB=$PWD/tools/variants/specstats.so timeout 300 python tools/specstats.py cfg3 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_spec_verification_rates.txt
for s in 1 0; do echo "MVX_SPEC=$s"; MVX_SPEC=$s timeout 600 python bench.py --no-cpu --no-traffic --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-1200; done | tee gpurun_out/r4_spec_first_bench.txt
