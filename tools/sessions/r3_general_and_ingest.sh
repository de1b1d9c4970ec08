#!/bin/bash
# round 3: the ingest-inclusive secondary metric, a counter pass of the GENERAL search kernel (MVX_GENERAL=1: what every non-default search,
# SATD and the two-per-SIMD fall-back run), and shell experiments (search LDS floor, request threads)
export TMPDIR=/tmp
out=$PWD/gpurun_out
timeout 600 python bench.py --ingest --no-cpu --no-traffic --steps 3 2>&1 | tail -1 > $out/r3_bench_ingest.json; python -c "
import json; d=json.loads(open('$out/r3_bench_ingest.json').read()); print('resident', round(d['value'],1), 'fps; ingest-inclusive', d.get('ingest_inclusive'), 'parity', d['parity_check']['identical'])"
MVX_GENERAL=1 timeout 600 python bench.py --no-cpu --no-traffic --batch 336 --steps 2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('general kernel, batch 336 (2016 chains):', round(d['value'],1), 'fps', round(d['roofline']['avg_launch_ms'],1), 'ms/launch parity', d['parity_check']['identical'])" | tee $out/r3_general_kernel.txt
MVX_GENERAL=1 BENCH_ARGS="--batch 336" bash tools/gpu_session.sh sq > /dev/null 2>&1
grep "analyse_kernel" $out/pmc_summary.txt >> $out/r3_general_kernel.txt; grep -c analyse_kernel $out/r3_general_kernel.txt
for cfg in "0 32" "54000 32" "0 16" "54000 64"; do set -- $cfg; echo "search LDS floor $1 request threads $2"; MVX_VS_SEARCH_LDS=$1 VS_NOVERIFY=1 python tools/vs_4k_run.py 640 $2 2>&1 | grep -E "steady|second half|thread-seconds"; done 2>&1 | tee $out/r3_vs_search_lds_threads.txt
