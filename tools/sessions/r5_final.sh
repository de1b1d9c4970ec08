#!/bin/bash
# r5 final state: the whole GPU suite, the driver's bench command (20 steps like the driver), rocprofv3 kernel stats of the same command, per-kernel HBM traffic,
# SQ / TCP / TCC counters of the search kernel
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r5_tests_gpu_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/r5_smoke.txt
t0=$(date +%s)
timeout 1500 python bench.py --steps 20 --warmup 5 > $out/r5_bench_default.json 2> $out/r5_bench_default.err || tail -5 $out/r5_bench_default.err
echo "bench.py --steps 20 --warmup 5 wall: $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('default', round(d['value'],1), 'fps', round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step frac', round(r['frac'],4), 'alone', r.get('launch_alone'), 'traffic', r['traffic'], r['traffic_source'][:120], 'parity', d['parity_check']['identical'], 'cpu', d['cpu_baseline']['value'])
PY
TAG=r5 BENCH_ARGS="--no-others --no-vs" bash tools/gpu_session.sh stats traffic > $out/r5_profile_steps.log 2>&1
PMC_FILTER=analyse_spec bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python bench.py --no-cpu --no-parity --no-traffic --no-others --steps 1 --warmup 0 --slots 1 > /dev/null 2>&1
cp $out/pmc_summary.txt $out/r5_search_final_counters.txt
head -14 $out/r5_kernel_stats.csv | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_pmc_traffic.json'))
for k,v in d['kernels'].items(): print(k[:70], round(v['hbm_bytes_per_dispatch_corrected']/1e9,1), 'GB per dispatch', v['dispatches'])
PY
cat $out/r5_search_final_counters.txt | cut -c1-160
