#!/bin/bash
# r5: row passes for 32x32 blocks (cfg5), the r4 build that "disagreed with the oracle on one 8K bench clip": does it still, and where
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:60], 'parity', d.get('parity_check',{}))"; }
export MVX_LIB=$PWD/tools/variants/strip32.so
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "full_size_parity_cfg5 or team_sizes or analyse_parity or properties_cfg5" 2>&1 | tail -15 | tee $out/r5_strip32_tests.txt
{
timeout 600 python bench.py --config cfg5 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg5 row passes (32x32)"
MVX_TEAM=0 MVX_LIB= timeout 600 python bench.py --config cfg5 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg5 product build (serial lean kernel)"
} 2>&1 | tee $out/r5_strip32_bench.txt
