#!/bin/bash
# round 2, final GPU session: the whole GPU suite, smoke(), then the profile round (bench line, kernel stats, HBM traffic)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $out/final_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $out/final_smoke.txt
bash tools/profile_round.sh r2 2>&1 | tail -12
for c in cfg5 cfg2 cfg4 cfg1; do timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 --config $c 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', round(d['value'],1), d['unit'], round(d['roofline']['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step')"; done 2>&1 | tee $out/final_other_configs.txt
