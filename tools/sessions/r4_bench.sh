#!/bin/bash
# round 4: the default bench line (CPU leg + parity, traffic passes, other_configs), timed
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
S=$(date +%s)
timeout 1500 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "rc $? wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_bench_default.json') if l.startswith('{')][-1])
print(round(d['value'],1), d['unit'], 'parity', d['parity_check']['identical'], 'traffic', d['roofline']['traffic'], 'frac', round(d['roofline']['frac'],4))
print(json.dumps(d.get('other_configs'), indent=1)[:2400])
print(d['cpu_baseline'].get('reference_ratio_estimate'))
PY
