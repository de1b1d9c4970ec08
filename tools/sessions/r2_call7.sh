#!/bin/bash
# round 2, GPU call 7: concurrent-request shell test, cfg4 full-size test and bench
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_vs_shim.py tests/test_gpu_parity.py -x -q -m gpu -k "concurrent or cfg4 or shim" 2>&1 | tail -15 | tee $out/c7_tests.txt
timeout 600 python bench.py --config cfg4 --steps 2 --warmup 1 2>&1 | tail -3 | tee $out/c7_cfg4.json
timeout 300 python bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu --batch 1023 2>&1 | tail -1 | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -o kt -- python $OLDPWD/bench.py --config cfg4 --no-cpu --steps 2 --warmup 1 > /tmp/kt4.log 2>&1)
python3 - <<'PY'
import glob, shutil
f = glob.glob('/tmp/kt4/**/*kernel_stats.csv', recursive=True)
if f: shutil.copy(f[0], 'gpurun_out/c7_cfg4_kernel_stats.csv'); print(open(f[0]).read()[:3000])
PY
