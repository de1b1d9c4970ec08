#!/bin/bash
# The GPU sessions of round 6, one function per session.
#     gpurun --timeout 1500 -- 'bash tools/sessions/r6.sh <session>'
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', 'alone', (r.get('launch_alone') or {}).get('ms'), r['kernel'][:60], 'parity', d.get('parity_check',{}).get('identical'))"; }

r6_formats() {
    # r6 first session: the new 4:4:4 / 4:2:2 / Gray cases and the overlap-0 Degrain cases, then the starting point of the round (unchanged library)
    timeout 900 python -m pytest tests/test_gpu_formats.py -q -m gpu 2>&1 | tail -40 | tee $out/r6_formats_tests.txt
    timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "degrain_parity" 2>&1 | tail -8 | tee $out/r6_degrain_side_tests.txt
    timeout 600 python bench.py --no-cpu --no-traffic --no-others --no-vs --steps 5 --warmup 2 2>&1 | tail -1 | tee $out/r6_start_bench.json | line "cfg3 start of r6"
}

"r6_$1" "${@:2}"
