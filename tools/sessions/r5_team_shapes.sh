#!/bin/bash
# r5: the team form for shapes WITHOUT row passes (one block at a time in phase A): small launches of cfg5 (32x32) and of 8-bit 16x16 blocks
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:52], 'parity', d.get('parity_check',{}).get('identical'))"; }
{
for cb in "cfg5 11" "cfg5 43" "hd16 64" "hd16 256" "hd16 512"; do set -- $cb
  timeout 400 python bench.py --config $1 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $2 2>&1 | tail -1 | line "$1 batch $2 library default"
  for t in 4 8; do MVX_SPEC=5 MVX_TEAM=$t timeout 400 python bench.py --config $1 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $2 2>&1 | tail -1 | line "$1 batch $2 speculative kernel, teams of $t"; done
done
timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 batch 2048 library default"
MVX_SPEC=5 MVX_TEAM=0 timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 batch 2048 speculative kernel, one wave per chain"
} 2>&1 | tee $out/r5_team_other_shapes.txt
