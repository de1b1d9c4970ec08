#!/bin/bash
# round 2, GPU call 10: barrier interval / chains-per-SIMD re-check with the UV plane, other configs, SQ counters of the final kernel
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")" || echo "$name FAILED"; }
{
r cfg3-sync64 MVX_CPW_SYNC=64 --
r cfg3-sync128 MVX_CPW_SYNC=128 --
r cfg3-sync256 MVX_CPW_SYNC=256 --
r cfg3-k4-b672 X=1 -- --batch 672
r cfg3-k3-b672 MVX_FAST_WPE=3 -- --batch 672
r cfg3-noxcd MVX_FAST_FLAGS=0 --
r cfg2 X=1 -- --config cfg2
r cfg4 X=1 -- --config cfg4
r cfg1 X=1 -- --config cfg1
} 2>&1 | tee $out/c10_variants.txt
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  -- python bench.py --no-cpu --steps 1 --warmup 0 > $out/c10_pmc.log 2>&1
cp $out/pmc_summary.txt $out/c10_pmc_summary.txt; grep -E "analyse_fast|group" $out/c10_pmc_summary.txt
