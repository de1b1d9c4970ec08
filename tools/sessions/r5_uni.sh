#!/bin/bash
# r5: uniform windows in the row passes (stage 1 in strip form with the pattern riding along): parity, then A/B against MVX_SPEC_UNI=0
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:60], 'parity', d.get('parity_check',{}).get('identical'))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or team or full_size or analyse_parity or golden" 2>&1 | tail -8 | tee $out/r5_uni_tests.txt
if grep -q "failed\|error" $out/r5_uni_tests.txt; then echo "not green: no timing"; exit 1; fi
{
for c in cfg3 cfg2 cfg4; do for u in 1 0; do
  MVX_SPEC_UNI=$u timeout 300 python bench.py --config $c --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "$c uniform windows $u"
done; done
} 2>&1 | tee $out/r5_uniform_windows_ab.txt
