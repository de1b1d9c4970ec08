#!/bin/bash
# r5: re-pricing the shifted luma copy (shadow plane 0) against the row-pass kernel: search + Degrain reading it / not reading it
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', 'parity', d.get('parity_check',{}).get('identical'))"; }
{
timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 default (search and Degrain read odd-sample blocks from the shifted luma copy)"
MVX_SHADOW_PLANES=2 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 search reads the plain luma plane (UV plane kept)"
MVX_SHADOW_PLANES=2 MVX_DEGRAIN_SHADOW=0 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 search AND Degrain read the plain luma plane"
MVX_DEGRAIN_SHADOW=0 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 only Degrain reads the plain luma plane"
MVX_SHADOW_PLANES=2 timeout 300 python bench.py --config cfg5 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg5 search reads the plain luma plane"
} 2>&1 | tee $out/r5_shadow_ab.txt
