#!/bin/bash
# round 4, final state: SQ / TA / TCP / TCC counters of the search kernel (cfg3 default batch)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
export PMC_FILTER="analyse_spec"
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" -- python bench.py --no-cpu --no-parity --no-traffic --no-others --steps 1 --warmup 0 > /dev/null 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r4_search_final_counters.txt
