#!/bin/bash
# r5 call 1: the whole GPU suite on the commit with the stream-ordering fix, the sharding case 30x in one process, a short bench line
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $out/r5_tests_gpu_first.txt
timeout 600 python tools/stress_sharding.py 30 2>&1 | tail -8 | tee $out/r5_sharding_stress_30x.txt
timeout 600 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 > $out/r5_bench_first.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_first.json'))
print('bench', round(d['value'],1), 'fps', round(d['roofline']['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))
PY
