#!/bin/bash
# round 4: full GPU suite after the ADVICE fixes; the default bench line with other_configs (timed); three chains per SIMD with 8 / 12 row loads in flight
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r4_tests_gpu.txt
O=gpurun_out/r4_spec_k3.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'), d['roofline']['kernel'][:44])" >> $O; }
EXTRA="--batch 512"; run "3072 chains, three per SIMD, 12 row loads in flight" MVX_FAST_K=3
EXTRA="--batch 512"; run "3072 chains, three per SIMD, 8 row loads in flight" MVX_FAST_K=3 MVX_LIB=$PWD/tools/variants/sw8.so
cat $O
/usr/bin/time -v timeout 1200 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; grep -E "Elapsed|Maximum resident" gpurun_out/r4_bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_bench_default.json') if l.startswith('{')][-1])
print(round(d['value'],1), d['unit'], 'parity', d['parity_check']['identical'], 'traffic', d['roofline']['traffic'], 'frac', round(d['roofline']['frac'],4))
print(json.dumps(d.get('other_configs'), indent=1)[:1800])
print(d['cpu_baseline'].get('reference_ratio_estimate'))
PY
