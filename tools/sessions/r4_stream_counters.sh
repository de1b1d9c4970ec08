#!/bin/bash
# round 4: what bounds the streaming kernels (Degrain cell kernel, Super row kernels): SQ / TA / TCP counters per dispatch
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
export PMC_FILTER="degrain_cell|super_rows|super_shadow_kernel|super_reduce_rows"
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python bench.py --no-cpu --no-parity --no-traffic --no-others --steps 1 --warmup 0 > /dev/null 2>&1
cp gpurun_out/pmc_summary.txt gpurun_out/r4_stream_kernel_counters.txt
