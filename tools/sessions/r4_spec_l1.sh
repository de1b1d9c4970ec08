#!/bin/bash
# round 4, session 5: is the streamed pass A bound by L1 misses?  one reference frame per CU (6-chain workgroups, one per CU) against two; counters
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_spec_stream_l1.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --no-parity --steps 2 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch')" >> $O; }
run "6 chains per workgroup (one reference frame), two workgroups per CU" MVX_FAST_CPW=6
run "6 chains per workgroup, ONE workgroup per CU (LDS floor 90 KB)" MVX_FAST_CPW=6 MVX_FAST_LDS_MIN=92160
run "12 chains per workgroup, one workgroup per CU (default)" A=1
run "4 chains per workgroup, 3 per CU" MVX_FAST_CPW=4
run "6 chains per workgroup, one per CU, two-per-SIMD build (MVX_FAST_K=2)" MVX_FAST_CPW=6 MVX_FAST_LDS_MIN=92160 MVX_FAST_K=2
cat $O
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python bench.py --no-cpu --no-parity --no-traffic --steps 1 --warmup 0 > /dev/null 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r4_spec_stream_kernel_counters.txt; grep analyse_spec gpurun_out/pmc_summary.txt | cut -c56-160
