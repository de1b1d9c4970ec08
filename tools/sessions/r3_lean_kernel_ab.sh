#!/bin/bash
# (record of a finished series: the -D switches these variant builds used were removed from mvx_analyse_fast.h once the winners were in; results in profiles/r3_lean_kernel_*_ab.txt)
# lean search kernel, round-3 instruction / latency trims: parity of the default build, then the same bench with each variant library
# (tools/build_variant.py: MVX_GSUM2 / MVX_PRED_LANES / MVX_SRC_AHEAD), then the default bench line with its checks
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r3_lean_kernel_ab.txt; : > $O
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step')"; }
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/tests_r3.txt
for v in old gsum ahead2 w12 w12a3; do
  [ -f tools/variants/$v.so ] && MVX_LIB=$PWD/tools/variants/$v.so timeout 300 python bench.py --no-cpu --no-traffic --steps 3 --warmup 1 2>&1 | tail -1 | line $v | tee -a $O
done
timeout 300 python bench.py --no-cpu --no-traffic --steps 3 --warmup 1 2>&1 | tail -1 | line default_build | tee -a $O
timeout 900 python bench.py > gpurun_out/bench_r3.json 2> gpurun_out/bench_r3.err || tail -5 gpurun_out/bench_r3.err
head -c 2500 gpurun_out/bench_r3.json
