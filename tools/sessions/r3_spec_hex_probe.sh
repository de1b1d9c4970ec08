#!/bin/bash
# (record: the MVX_SPEC_HEX code this probe ran lives in commit afd59f1 only -- it measured slower and was removed)
# the fused predictor + hexagon pass (MVX_SPEC_HEX=1, tools/variants/spec_hex.so), as far as the last 48 seconds of the round's GPU time go:
# the default bench with it, then the search parity cases (every Analyse configuration + the full-size 4K16 byte parity)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp MVX_LIB=$PWD/tools/variants/spec_hex.so
mkdir -p gpurun_out
O=gpurun_out/r3_spec_hex_probe.txt
echo "MVX_SPEC_HEX=1 build" > $O
timeout 25 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('bench:', round(d['value'],1), d['unit'], round(d['roofline']['avg_launch_ms'],1), 'ms per search launch', round(d['ms_per_step'],1), 'ms per step')" >> $O 2>&1
cat $O
timeout 30 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "analyse_parity or full_size_parity_cfg3" 2>&1 | tail -3 >> $O
cat $O
