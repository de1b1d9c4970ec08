#!/bin/bash
# r5: the team form of the speculative kernel (nw waves walk one chain): parity first, then ms per launch against the one-wave form
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:60], 'parity', d.get('parity_check',{}).get('identical'))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team" 2>&1 | tail -8 | tee $out/r5_team_tests.txt
if grep -q "failed\|error\|Timeout" $out/r5_team_tests.txt; then echo "team tests not green: no timing"; exit 1; fi
{
for t in 0 4 2 8 3; do
  MVX_TEAM=$t timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg3 batch 341 team $t"
done
for b in 86 22; do for t in 0 4 8; do
  MVX_TEAM=$t timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $b 2>&1 | tail -1 | line "cfg3 batch $b team $t"
done; done
} 2>&1 | tee $out/r5_team_bench.txt
