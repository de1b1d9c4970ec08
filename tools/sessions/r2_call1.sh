#!/bin/bash
# round 2, GPU call 1: TA / L1 access-pattern microbenchmark, baseline bench, first variants, SQ / TA counters of the r1 kernel
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 120 tools/micro/ta_pattern > $out/ta_pattern.txt 2>&1; tail -25 $out/ta_pattern.txt
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")"; }
{
r base MVX_LIB=$PWD/tools/variants/base.so --
r x1 MVX_LIB=$PWD/tools/variants/x1.so --
r base-b512 MVX_LIB=$PWD/tools/variants/base.so -- --batch 512
r base-b512-wpe3 MVX_WPE3=1 MVX_LIB=$PWD/tools/variants/base.so -- --batch 512
r x1-b512-wpe3 MVX_WPE3=1 MVX_LIB=$PWD/tools/variants/x1.so -- --batch 512
r x1-b512 MVX_LIB=$PWD/tools/variants/x1.so -- --batch 512
} 2>&1 | tee $out/c1_variants.txt
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TA|TCP|TD|TCC|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u > $out/counters_gfx950.txt
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  -- python bench.py --no-cpu --steps 1 --warmup 0 > $out/c1_pmc.log 2>&1
cp $out/pmc_summary.txt $out/c1_pmc_summary.txt; cat $out/c1_pmc_summary.txt
