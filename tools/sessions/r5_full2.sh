#!/bin/bash
# r5: the whole GPU suite + the driver's bench command on the current tree
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r5_tests_gpu.txt
t0=$(date +%s)
timeout 1500 python bench.py > $out/r5_bench_default.json 2> $out/r5_bench_default.err || tail -5 $out/r5_bench_default.err
echo "bench.py wall: $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('default', round(d['value'],1), 'fps', round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step frac', round(r['frac'],4), 'traffic', r['traffic'], r['traffic_source'][:200], 'parity', d['parity_check']['identical'], 'cpu', d['cpu_baseline']['value'])
print(json.dumps(d.get('vs_shell'))[:1500])
print(json.dumps(d.get('other_configs'))[:2500])
PY
