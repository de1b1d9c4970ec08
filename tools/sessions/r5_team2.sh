#!/bin/bash
# r5: what the team form changes in the memory system (counters, team 4 against one wave per chain at the default batch), where a team wave waits,
# which team size pays at which launch size, and the round-4 stream-ordering race reproduced with the fix switched off
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:60], 'parity', d.get('parity_check',{}).get('identical'))"; }
echo "== stream-ordering race, fix switched off (MVX_BENCH_NO_STREAM_ORDER=1), then on" | tee $out/r5_sharding_race_repro.txt
MVX_BENCH_NO_STREAM_ORDER=1 timeout 300 python tools/stress_sharding.py 30 2>&1 | tail -6 | tee -a $out/r5_sharding_race_repro.txt
timeout 300 python tools/stress_sharding.py 30 2>&1 | tail -3 | tee -a $out/r5_sharding_race_repro.txt
{
for bt in "170 0" "170 2" "170 3" "128 0" "128 2" "128 3" "128 4" "43 4" "43 8" "43 0" "341 5" "341 6"; do set -- $bt
  MVX_TEAM=$2 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $1 2>&1 | tail -1 | line "cfg3 batch $1 team $2"
done
} 2>&1 | tee $out/r5_team_batch_sweep.txt
for t in 4 0; do
  echo "== where a wave's time goes, team $t (instrumented build)" | tee -a $out/r5_team_phase_cycles.txt
  MVX_TEAM=$t MVX_LIB=$PWD/tools/variants/specprof.so timeout 300 python tools/specprof.py cfg3 341 2>&1 | tail -16 | tee -a $out/r5_team_phase_cycles.txt
done
for t in 4 0; do
  export MVX_TEAM=$t
  TAG=r5_team$t BENCH_ARGS="--no-others" bash tools/gpu_session.sh traffic > /dev/null 2>&1
  PMC_FILTER=analyse_spec bash tools/pmc.sh "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" -- python bench.py --no-cpu --no-parity --no-traffic --no-others --steps 1 --warmup 0 > /dev/null 2>&1
  cp $out/pmc_summary.txt $out/r5_team${t}_search_counters.txt
done
unset MVX_TEAM
python - <<'PY'
import json
for t in (4, 0):
    d = json.load(open('gpurun_out/r5_team%d_pmc_traffic.json' % t))
    for k, v in d['kernels'].items():
        if 'analyse_spec' in k: print('team', t, k[:50], 'HBM GB per launch', round(v['hbm_bytes_per_dispatch_corrected'] / 1e9, 1))
PY
grep -h "TCC_\|TCP_TCC\|TA_BUSY\|SQ_INSTS_VMEM\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES " $out/r5_team4_search_counters.txt $out/r5_team0_search_counters.txt | cut -c1-200
