#!/bin/bash
# r5: two batches in flight with the searches chained (one launch at a time), Super / Degrain of the neighbouring batches under the running search
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', 'parity', d.get('parity_check',{}).get('identical'))"; }
{
timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 1 2>&1 | tail -1 | line "cfg3 one batch in flight"
timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 2 2>&1 | tail -1 | line "cfg3 two batches in flight, searches chained"
timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 9 --warmup 3 --slots 3 --batch 256 2>&1 | tail -1 | line "cfg3 three batches of 256 in flight, searches chained"
timeout 400 python bench.py --config cfg2 --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 2 2>&1 | tail -1 | line "cfg2 two batches in flight"
timeout 400 python bench.py --config cfg2 --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 1 2>&1 | tail -1 | line "cfg2 one batch in flight"
} 2>&1 | tee $out/r5_batches_in_flight.txt
