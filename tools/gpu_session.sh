#!/bin/bash
# One GPU session = a list of steps, run in order on the MI355X box from the repository root:
#     gpurun --timeout 900 -- 'bash tools/gpu_session.sh tests bench'
# Every step writes under gpurun_out/ (merged back by gpurun); copy what DESIGN.md quotes into profiles/.
# Steps:  tests [pytest -k expr]   the whole -m gpu suite (or TESTS_K="expr")
#         smoke                    __graft_entry__.smoke()
#         bench                    default bench line (BENCH_ARGS="..." for other arguments) -> gpurun_out/bench_<tag>.json
#         configs                  one short bench line per other configuration
#         stats                    rocprofv3 --kernel-trace --stats of the default bench -> gpurun_out/<tag>_kernel_stats.csv
#         traffic                  FETCH_SIZE / WRITE_SIZE passes for every kernel -> gpurun_out/<tag>_pmc_traffic.json
#         sq                       SQ / TCP counter passes of the search kernel -> gpurun_out/<tag>_search_sq_counters.txt
#         micro <name>             build and run tools/micro/<name>.hip
#         ab <libA> <libB> ...     bench --no-cpu with MVX_LIB=<lib> for each named library build (tools/variants/*.so)
#         vs                       the VapourSynth shell on a 4K16 clip (tools/vs_4k_run.py)
#         sh <file>                a one-off script
export TMPDIR=/tmp
TAG=${TAG:-r5}
out=$PWD/gpurun_out; mkdir -p $out
root=$PWD
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', 'frac', round(r['frac'],4), 'parity', d.get('parity_check',{}).get('identical'))"; }
while [ $# -gt 0 ]; do
  step=$1; shift
  case $step in
    tests)
      timeout 1700 python -m pytest tests -x -q -m gpu ${TESTS_K:+-k "$TESTS_K"} 2>&1 | tail -15 | tee $out/tests_$TAG.txt ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke_$TAG.txt ;;
    bench)
      timeout 900 python bench.py $BENCH_ARGS > $out/bench_$TAG.json 2> $out/bench_$TAG.err || tail -5 $out/bench_$TAG.err
      cat $out/bench_$TAG.json | line default; head -c 3000 $out/bench_$TAG.json ;;
    configs)
      for c in cfg5 cfg2 cfg4 cfg1; do timeout 400 python bench.py --no-cpu --no-traffic --steps 2 --warmup 1 --config $c 2>&1 | tail -1 | line $c; done 2>&1 | tee $out/configs_$TAG.txt ;;
    stats)
      (cd /tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $root/bench.py --no-cpu --no-parity --no-traffic $BENCH_ARGS > /tmp/kt.log 2>&1)
      f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/${TAG}_kernel_stats.csv && head -12 $out/${TAG}_kernel_stats.csv || tail -5 /tmp/kt.log ;;
    traffic)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && rm -rf /tmp/pmc_$c && timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "${PMC_KERNELS:-analyse_|degrain|super_|compensate|blockfps|bf_|usable}" --output-format csv -d /tmp/pmc_$c -o p -- python $root/bench.py --no-cpu --no-parity --no-traffic --steps 1 --warmup 0 $BENCH_ARGS > /tmp/pmc_$c.log 2>&1)
      done
      python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "$BENCH_ARGS" > $out/${TAG}_pmc_traffic.json; head -c 1200 $out/${TAG}_pmc_traffic.json ;;
    sq)
      bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" -- python bench.py --no-cpu --no-parity --no-traffic --steps 1 --warmup 0 $BENCH_ARGS > /dev/null 2>&1
      cp $out/pmc_summary.txt $out/${TAG}_search_sq_counters.txt; grep analyse_fast $out/pmc_summary.txt | head -40 ;;
    micro)
      n=$1; shift
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_$n tools/micro/$n.hip && timeout 300 /tmp/micro_$n 2>&1 | tee $out/${TAG}_micro_$n.txt ;;
    ab)
      : > $out/ab_$TAG.txt
      while [ $# -gt 0 ] && [ -f "$1" ]; do
        lib=$1; shift
        MVX_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu --no-traffic $BENCH_ARGS 2>&1 | tail -1 | line $lib | tee -a $out/ab_$TAG.txt
      done ;;
    vs)
      timeout 900 python tools/vs_4k_run.py $VS_ARGS 2>&1 | tail -25 | tee $out/${TAG}_vs_shell.txt ;;
    sh)
      f=$1; shift; bash $f 2>&1 | tail -40 ;;
    *) echo "unknown step $step" ;;
  esac
done
