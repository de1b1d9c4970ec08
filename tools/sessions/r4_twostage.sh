#!/bin/bash
# round 4: two-stage row passes (predictor pass, then the pattern around the predictor phase's winner)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "analyse or golden or full_size_parity_cfg3" 2>&1 | tail -5 | tee gpurun_out/r4_twostage_tests.txt
O=gpurun_out/r4_spec_twostage.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))" >> $O; }
EXTRA=""; run "default (batch 341)" A=1
cat $O
MVX_LIB=$PWD/tools/variants/specprof.so timeout 200 python tools/specprof.py cfg3 341 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_spec_phase_cycles_twostage.txt
MVX_LIB=$PWD/tools/variants/specstats.so timeout 300 python tools/specstats.py cfg3 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_spec_twostage_verification_rates.txt
