#!/bin/bash
# round 4, session 3: what bounds pass A of the speculative kernel -- counters, chains per SIMD, three timing-only ablations
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_spec_diag.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" timeout 400 python bench.py --no-cpu --no-traffic --no-parity --steps 2 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value'],1),'fps', round(d['roofline']['avg_launch_ms'],1),'ms/launch')" >> $O; }
run "default (3 per SIMD)" A=1
run "two chains per SIMD (MVX_FAST_K=2)" MVX_FAST_K=2
run "one chain per SIMD (MVX_FAST_K=1)" MVX_FAST_K=1
run "abl1: no zero/global/hier pass" MVX_LIB=$PWD/tools/variants/abl1.so
run "abl2: all 16 groups of the pattern pass read the centre block" MVX_LIB=$PWD/tools/variants/abl2.so
run "abl3: every speculative result taken (no live blocks)" MVX_LIB=$PWD/tools/variants/abl3.so
run "barrier off (MVX_CPW_SYNC=0)" MVX_CPW_SYNC=0
cat $O
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python bench.py --no-cpu --no-parity --no-traffic --steps 1 --warmup 0 > /dev/null 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r4_spec_kernel_counters.txt; grep analyse_spec gpurun_out/pmc_summary.txt | cut -c1-200
