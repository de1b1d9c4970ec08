#!/bin/bash
# round 4: the whole GPU suite, the default bench line (CPU leg, traffic passes, parity check), the FETCH_SIZE calibration microkernel
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r4_tests_gpu.txt
timeout 900 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err || tail -5 gpurun_out/r4_bench_default.err
head -c 2500 gpurun_out/r4_bench_default.json; echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_fetch_calib tools/micro/fetch_calib.hip 2>/dev/null
(cd /tmp && rm -rf /tmp/fc && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -o p -- /tmp/micro_fetch_calib > /tmp/fc.log 2>&1; grep pattern /tmp/fc.log)
python3 - <<'PY' | tee gpurun_out/r4_fetch_size_calibration.txt
import csv, glob
rows = []
for f in glob.glob('/tmp/fc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            rows.append((r['Kernel_Name'][:40], float(r['Counter_Value'])))
print(open('/tmp/fc.log').read().strip().split('\n')[-3:])
for k, v in rows:
    print('%-40s FETCH_SIZE %.0f (x 1024 = %.3f GB; region 12.885 GB -> bytes per count-KB %.3f)' % (k, v, v * 1024 / 1e9, 12.884901888e9 / (v * 1024)))
PY
