#!/bin/bash
# r5: column passes (stage 2 of strip-form windows without re-loading rows): parity, then A/B against MVX_SPEC_COL=0
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:40], 'parity', d.get('parity_check',{}).get('identical'))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative_kernel and not 8bit or team or full_size_parity_cfg3 or analyse_parity or golden" 2>&1 | tail -8 | tee $out/r5_col_tests.txt
if grep -q "failed\|error" $out/r5_col_tests.txt; then echo "not green: no timing"; exit 1; fi
{
for u in 1 0; do
  MVX_SPEC_COL=$u timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 column passes $u"
done
MVX_SPEC_COL=1 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 64 2>&1 | tail -1 | line "cfg3 batch 64 (team) column passes 1"
MVX_SPEC_COL=0 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 64 2>&1 | tail -1 | line "cfg3 batch 64 (team) column passes 0"
} 2>&1 | tee $out/r5_column_passes_ab.txt
