#!/bin/bash
# (record of a finished series: the -D switches these variant builds used were removed from mvx_analyse_fast.h once the winners were in; results in profiles/r3_lean_kernel_*_ab.txt)
# lean search kernel: predictor de-duplication and the hexagon pass without the speculative square (less L1 traffic, one pass more)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r3_lean_kernel_ab2.txt; : > $O
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step')"; }
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/tests_r3.txt
timeout 300 python bench.py --no-cpu --no-traffic --steps 3 --warmup 1 2>&1 | tail -1 | line "default_build(dedup,no_spec)" | tee -a $O
for v in spec_dedup nospec_nodedup w12; do
  [ -f tools/variants/$v.so ] && MVX_LIB=$PWD/tools/variants/$v.so timeout 300 python bench.py --no-cpu --no-traffic --steps 3 --warmup 1 2>&1 | tail -1 | line $v | tee -a $O
done
MVX_LIB=$PWD/tools/variants/spec_dedup.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "analyse or full_size or golden" 2>&1 | tail -2 | tee -a $O
