#!/bin/bash
# r5: the whole GPU suite (team form as the library's choice for small launches), the driver's bench command (with the vs_shell leg), cfg2 team sweep
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:60], 'parity', d.get('parity_check',{}).get('identical'))"; }
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r5_tests_gpu_team_default.txt
t0=$(date +%s)
timeout 1500 python bench.py > $out/r5_bench_default.json 2> $out/r5_bench_default.err || tail -5 $out/r5_bench_default.err
echo "bench.py wall: $(( $(date +%s) - t0 )) s"
cat $out/r5_bench_default.json | line default
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
print(json.dumps(d.get('vs_shell'), indent=1)[:2500])
print(json.dumps(d.get('other_configs'))[:1500])
PY
{
for bt in "1024 0" "512 0" "512 2" "256 0" "256 2" "256 4" "64 0" "64 4" "64 8"; do set -- $bt
  MVX_TEAM=$2 timeout 300 python bench.py --config cfg2 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $1 2>&1 | tail -1 | line "cfg2 batch $1 team $2"
done
} 2>&1 | tee $out/r5_team_cfg2_sweep.txt
