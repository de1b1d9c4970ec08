#!/bin/bash
# round 2, GPU call 3: whole GPU suite, SQ / TA counters of the lean kernel on shadow-layout frames, cfg2 diagnosis
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $out/c3_tests.txt
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r cfg2-lean-plain-k3 MVX_SHADOW=0 -- --config cfg2
r cfg2-lean-shadow-k2 MVX_FAST_WPE=2 -- --config cfg2
r cfg2-lean-plain-k2 MVX_SHADOW=0 MVX_FAST_WPE=2 -- --config cfg2
r cfg2-lean-shadow-k4 X=1 -- --config cfg2 --batch 2048
r cfg1-lean X=1 -- --config cfg1
r cfg1-general MVX_GENERAL=1 MVX_SHADOW=0 -- --config cfg1
} 2>&1 | tee $out/c3_variants.txt
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" \
  "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  -- python bench.py --no-cpu --steps 1 --warmup 0 > $out/c3_pmc.log 2>&1
cp $out/pmc_summary.txt $out/c3_pmc_summary.txt; grep -E "analyse_fast|group" $out/c3_pmc_summary.txt
