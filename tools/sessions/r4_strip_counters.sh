#!/bin/bash
# round 4, session 9: counters of the row-pass kernel at two chains per SIMD (the build without spills), with and without runs
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "1" "3"; do
MVX_SPEC=$v MVX_FAST_K=2 bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python bench.py --no-cpu --no-parity --no-traffic --steps 1 --warmup 0 > /dev/null 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r4_strip_counters_spec$v.txt; echo "MVX_SPEC=$v MVX_FAST_K=2"; grep analyse_spec gpurun_out/pmc_summary.txt | cut -c56-150
done
