#!/bin/bash
# r5: 8-bit 16x16 blocks overlapping by 8 through the row passes (8-byte columns): parity, then the bench line against the serial kernel
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:52], 'parity', d.get('parity_check',{}).get('identical'))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "8bit_16x16 or speculative or analyse_parity or golden or full_size_parity_cfg2" 2>&1 | tail -8 | tee $out/r5_hd16_tests.txt
if grep -q "failed\|error" $out/r5_hd16_tests.txt; then echo "not green: no timing"; exit 1; fi
{
timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 row passes, barrier every 128 blocks"
MVX_CPW_SYNC=0 timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 row passes, no barrier"
MVX_SPEC=0 timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 serial lean kernel"
timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 1024 2>&1 | tail -1 | line "hd16 batch 1024 (2048 chains) row passes"
timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 128 2>&1 | tail -1 | line "hd16 batch 128 (teams)"
} 2>&1 | tee $out/r5_hd16_bench.txt
