#!/bin/bash
# The GPU sessions of round 5, one function per session (the r3 / r4 one-off scripts this directory used to hold are in the git history:
# `git log -- tools/sessions`; profiles/README.md names the session behind every r5 file).
#     gpurun --timeout 1500 -- 'bash tools/sessions/r5.sh <session>'        sessions: col final first full full2 hd16 shadow_ab slots strip32 team team2 team_shapes uni vs_trace
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', r['kernel'][:60], 'parity', d.get('parity_check',{}).get('identical'))"; }

r5_col() {
    # r5: column passes (stage 2 of strip-form windows without re-loading rows): parity, then A/B against MVX_SPEC_COL=0
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative_kernel and not 8bit or team or full_size_parity_cfg3 or analyse_parity or golden" 2>&1 | tail -8 | tee $out/r5_col_tests.txt
if grep -q "failed\|error" $out/r5_col_tests.txt; then echo "not green: no timing"; exit 1; fi
    {
for u in 1 0; do
      MVX_SPEC_COL=$u timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 column passes $u"
    done
    MVX_SPEC_COL=1 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 64 2>&1 | tail -1 | line "cfg3 batch 64 (team) column passes 1"
    MVX_SPEC_COL=0 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 64 2>&1 | tail -1 | line "cfg3 batch 64 (team) column passes 0"
    } 2>&1 | tee $out/r5_column_passes_ab.txt
}

r5_final_light() {
    # r5: the whole GPU suite, smoke() and the driver's bench command on the last commit that touches bench.py (the profiler passes of `final` are not repeated:
    # the library is unchanged)
    timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r5_tests_gpu_final.txt
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/r5_smoke.txt
    t0=$(date +%s)
    timeout 1500 python bench.py --steps 20 --warmup 5 > $out/r5_bench_default.json 2> $out/r5_bench_default.err || tail -5 $out/r5_bench_default.err
    echo "bench.py --steps 20 --warmup 5 wall: $(( $(date +%s) - t0 )) s"
    tail -1 $out/r5_bench_default.json | cut -c1-300
}

r5_final() {
    # r5 final state: the whole GPU suite, the driver's bench command (20 steps like the driver), rocprofv3 kernel stats of the same command, per-kernel HBM traffic,
    # SQ / TCP / TCC counters of the search kernel
    timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r5_tests_gpu_final.txt
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/r5_smoke.txt
    t0=$(date +%s)
    timeout 1500 python bench.py --steps 20 --warmup 5 > $out/r5_bench_default.json 2> $out/r5_bench_default.err || tail -5 $out/r5_bench_default.err
    echo "bench.py --steps 20 --warmup 5 wall: $(( $(date +%s) - t0 )) s"
    python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('default', round(d['value'],1), 'fps', round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step frac', round(r['frac'],4), 'alone', r.get('launch_alone'), 'traffic', r['traffic'], r['traffic_source'][:120], 'parity', d['parity_check']['identical'], 'cpu', d['cpu_baseline']['value'])
PY
    TAG=r5 BENCH_ARGS="--no-others --no-vs" bash tools/gpu_session.sh stats traffic > $out/r5_profile_steps.log 2>&1
    PMC_FILTER=analyse_spec bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python bench.py --no-cpu --no-parity --no-traffic --no-others --steps 1 --warmup 0 --slots 1 > /dev/null 2>&1
    cp $out/pmc_summary.txt $out/r5_search_final_counters.txt
    head -14 $out/r5_kernel_stats.csv | cut -c1-200
    python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_pmc_traffic.json'))
for k,v in d['kernels'].items(): print(k[:70], round(v['hbm_bytes_per_dispatch_corrected']/1e9,1), 'GB per dispatch', v['dispatches'])
PY
    cat $out/r5_search_final_counters.txt | cut -c1-160
}

r5_first() {
    # r5 call 1: the whole GPU suite on the commit with the stream-ordering fix, the sharding case 30x in one process, a short bench line
    timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $out/r5_tests_gpu_first.txt
    timeout 600 python tools/stress_sharding.py 30 2>&1 | tail -8 | tee $out/r5_sharding_stress_30x.txt
    timeout 600 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 > $out/r5_bench_first.json
    python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_first.json'))
print('bench', round(d['value'],1), 'fps', round(d['roofline']['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step parity', d.get('parity_check',{}).get('identical'))
PY
}

r5_full() {
    # r5: the whole GPU suite (team form as the library's choice for small launches), the driver's bench command (with the vs_shell leg), cfg2 team sweep
    timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r5_tests_gpu_team_default.txt
    t0=$(date +%s)
    timeout 1500 python bench.py > $out/r5_bench_default.json 2> $out/r5_bench_default.err || tail -5 $out/r5_bench_default.err
    echo "bench.py wall: $(( $(date +%s) - t0 )) s"
    cat $out/r5_bench_default.json | line default
    python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
print(json.dumps(d.get('vs_shell'), indent=1)[:2500])
print(json.dumps(d.get('other_configs'))[:1500])
PY
    {
for bt in "1024 0" "512 0" "512 2" "256 0" "256 2" "256 4" "64 0" "64 4" "64 8"; do set -- $bt
      MVX_TEAM=$2 timeout 300 python bench.py --config cfg2 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $1 2>&1 | tail -1 | line "cfg2 batch $1 team $2"
    done
    } 2>&1 | tee $out/r5_team_cfg2_sweep.txt
}

r5_full2() {
    # r5: the whole GPU suite + the driver's bench command on the current tree
    timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r5_tests_gpu.txt
    t0=$(date +%s)
    timeout 1500 python bench.py > $out/r5_bench_default.json 2> $out/r5_bench_default.err || tail -5 $out/r5_bench_default.err
    echo "bench.py wall: $(( $(date +%s) - t0 )) s"
    python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('default', round(d['value'],1), 'fps', round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step frac', round(r['frac'],4), 'traffic', r['traffic'], r['traffic_source'][:200], 'parity', d['parity_check']['identical'], 'cpu', d['cpu_baseline']['value'])
print(json.dumps(d.get('vs_shell'))[:1500])
print(json.dumps(d.get('other_configs'))[:2500])
PY
}

r5_hd16() {
    # r5: 8-bit 16x16 blocks overlapping by 8 through the row passes (8-byte columns): parity, then the bench line against the serial kernel
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "8bit_16x16 or speculative or analyse_parity or golden or full_size_parity_cfg2" 2>&1 | tail -8 | tee $out/r5_hd16_tests.txt
if grep -q "failed\|error" $out/r5_hd16_tests.txt; then echo "not green: no timing"; exit 1; fi
    {
    timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 row passes, barrier every 128 blocks"
    MVX_CPW_SYNC=0 timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 row passes, no barrier"
    MVX_SPEC=0 timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 serial lean kernel"
    timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 1024 2>&1 | tail -1 | line "hd16 batch 1024 (2048 chains) row passes"
    timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch 128 2>&1 | tail -1 | line "hd16 batch 128 (teams)"
    } 2>&1 | tee $out/r5_hd16_bench.txt
}

r5_shadow_ab() {
    # r5: re-pricing the shifted luma copy (shadow plane 0) against the row-pass kernel: search + Degrain reading it / not reading it
    {
    timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 default (search and Degrain read odd-sample blocks from the shifted luma copy)"
    MVX_SHADOW_PLANES=2 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 search reads the plain luma plane (UV plane kept)"
    MVX_SHADOW_PLANES=2 MVX_DEGRAIN_SHADOW=0 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 search AND Degrain read the plain luma plane"
    MVX_DEGRAIN_SHADOW=0 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg3 only Degrain reads the plain luma plane"
    MVX_SHADOW_PLANES=2 timeout 300 python bench.py --config cfg5 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg5 search reads the plain luma plane"
    } 2>&1 | tee $out/r5_shadow_ab.txt
}

r5_slots() {
    # r5: batches in flight, each on its own stream: Super / Degrain of the neighbouring batches under the running search, and the next search launch in the wave slots
    # the current one frees.  (The first form of this chained the search launches by events -- profiles/r5_batches_in_flight_chained_searches.txt, 878 fps; not chaining
    # them is 922: profiles/r5_batches_in_flight_unchained.txt.)
    {
    timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 1 2>&1 | tail -1 | line "cfg3 one batch in flight"
    timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 2 2>&1 | tail -1 | line "cfg3 two batches in flight"
    timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 9 --warmup 3 --slots 3 --batch 256 2>&1 | tail -1 | line "cfg3 three batches of 256 in flight"
    timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 8 --warmup 4 --slots 4 --batch 170 2>&1 | tail -1 | line "cfg3 four batches of 170 in flight"
    timeout 400 python bench.py --config cfg2 --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 2 2>&1 | tail -1 | line "cfg2 two batches in flight"
    timeout 400 python bench.py --config cfg2 --no-cpu --no-traffic --no-others --steps 8 --warmup 2 --slots 1 2>&1 | tail -1 | line "cfg2 one batch in flight"
    } 2>&1 | tee $out/r5_batches_in_flight.txt
}

r5_strip32() {
    # r5: row passes for 32x32 blocks (cfg5), the r4 build that "disagreed with the oracle on one 8K bench clip": does it still, and where
    export MVX_LIB=$PWD/tools/variants/strip32.so
    timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "full_size_parity_cfg5 or team_sizes or analyse_parity or properties_cfg5" 2>&1 | tail -15 | tee $out/r5_strip32_tests.txt
    {
    timeout 600 python bench.py --config cfg5 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg5 row passes (32x32)"
    MVX_TEAM=0 MVX_LIB= timeout 600 python bench.py --config cfg5 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg5 product build (serial lean kernel)"
    } 2>&1 | tee $out/r5_strip32_bench.txt
}

r5_team() {
    # r5: the team form of the speculative kernel (nw waves walk one chain): parity first, then ms per launch against the one-wave form
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "team" 2>&1 | tail -8 | tee $out/r5_team_tests.txt
if grep -q "failed\|error\|Timeout" $out/r5_team_tests.txt; then echo "team tests not green: no timing"; exit 1; fi
    {
for t in 0 4 2 8 3; do
      MVX_TEAM=$t timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "cfg3 batch 341 team $t"
    done
for b in 86 22; do for t in 0 4 8; do
      MVX_TEAM=$t timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $b 2>&1 | tail -1 | line "cfg3 batch $b team $t"
    done; done
    } 2>&1 | tee $out/r5_team_bench.txt
}

r5_team2() {
    # r5: what the team form changes in the memory system (counters, team 4 against one wave per chain at the default batch), where a team wave waits,
    # which team size pays at which launch size, and the round-4 stream-ordering race reproduced with the fix switched off
    echo "== stream-ordering race, fix switched off (MVX_BENCH_NO_STREAM_ORDER=1), then on" | tee $out/r5_sharding_race_repro.txt
    MVX_BENCH_NO_STREAM_ORDER=1 timeout 300 python tools/stress_sharding.py 30 2>&1 | tail -6 | tee -a $out/r5_sharding_race_repro.txt
    timeout 300 python tools/stress_sharding.py 30 2>&1 | tail -3 | tee -a $out/r5_sharding_race_repro.txt
    {
for bt in "170 0" "170 2" "170 3" "128 0" "128 2" "128 3" "128 4" "43 4" "43 8" "43 0" "341 5" "341 6"; do set -- $bt
      MVX_TEAM=$2 timeout 300 python bench.py --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $1 2>&1 | tail -1 | line "cfg3 batch $1 team $2"
    done
    } 2>&1 | tee $out/r5_team_batch_sweep.txt
for t in 4 0; do
      echo "== where a wave's time goes, team $t (instrumented build)" | tee -a $out/r5_team_phase_cycles.txt
      MVX_TEAM=$t MVX_LIB=$PWD/tools/variants/specprof.so timeout 300 python tools/specprof.py cfg3 341 2>&1 | tail -16 | tee -a $out/r5_team_phase_cycles.txt
    done
for t in 4 0; do
      export MVX_TEAM=$t
      TAG=r5_team$t BENCH_ARGS="--no-others" bash tools/gpu_session.sh traffic > /dev/null 2>&1
      PMC_FILTER=analyse_spec bash tools/pmc.sh "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" -- python bench.py --no-cpu --no-parity --no-traffic --no-others --steps 1 --warmup 0 > /dev/null 2>&1
      cp $out/pmc_summary.txt $out/r5_team${t}_search_counters.txt
    done
    unset MVX_TEAM
    python - <<'PY'
import json
for t in (4, 0):
    d = json.load(open('gpurun_out/r5_team%d_pmc_traffic.json' % t))
    for k, v in d['kernels'].items():
        if 'analyse_spec' in k: print('team', t, k[:50], 'HBM GB per launch', round(v['hbm_bytes_per_dispatch_corrected'] / 1e9, 1))
PY
    grep -h "TCC_\|TCP_TCC\|TA_BUSY\|SQ_INSTS_VMEM\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES " $out/r5_team4_search_counters.txt $out/r5_team0_search_counters.txt | cut -c1-200
}

r5_team_shapes() {
    # r5: the team form for shapes WITHOUT row passes (one block at a time in phase A): small launches of cfg5 (32x32) and of 8-bit 16x16 blocks
    {
for cb in "cfg5 11" "cfg5 43" "hd16 64" "hd16 256" "hd16 512"; do set -- $cb
      timeout 400 python bench.py --config $1 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $2 2>&1 | tail -1 | line "$1 batch $2 library default"
      for t in 4 8; do MVX_SPEC=5 MVX_TEAM=$t timeout 400 python bench.py --config $1 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 --batch $2 2>&1 | tail -1 | line "$1 batch $2 speculative kernel, teams of $t"; done
    done
    timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 batch 2048 library default"
    MVX_SPEC=5 MVX_TEAM=0 timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 batch 2048 speculative kernel, one wave per chain"
    } 2>&1 | tee $out/r5_team_other_shapes.txt
}

r5_uni() {
    # r5: uniform windows in the row passes (stage 1 in strip form with the pattern riding along): parity, then A/B against MVX_SPEC_UNI=0
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or team or full_size or analyse_parity or golden" 2>&1 | tail -8 | tee $out/r5_uni_tests.txt
if grep -q "failed\|error" $out/r5_uni_tests.txt; then echo "not green: no timing"; exit 1; fi
    {
for c in cfg3 cfg2 cfg4; do for u in 1 0; do
      MVX_SPEC_UNI=$u timeout 300 python bench.py --config $c --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "$c uniform windows $u"
    done; done
    } 2>&1 | tee $out/r5_uniform_windows_ab.txt
}

r5_vs_trace() {
    # r5: where the shell's graph construction (3 s of a 5.9 s run) goes: window trace + thread-second statistics, 384 frames; team form off for comparison
    MVX_VS_KEEP_STDERR=$out/vs_team MVX_VS_STATS=1 MVX_VS_TRACE=1 timeout 600 python bench.py --vs-shell-leg --vs-frames 384 2>/dev/null | tail -1 > $out/r5_vs_leg_384_team.json
    MVX_TEAM=0 MVX_VS_KEEP_STDERR=$out/vs_noteam MVX_VS_STATS=1 timeout 600 python bench.py --vs-shell-leg --vs-frames 384 2>/dev/null | tail -1 > $out/r5_vs_leg_384_noteam.json
    python - <<'PY'
import json
for k in ('team', 'noteam'):
    d = json.load(open('gpurun_out/r5_vs_leg_384_%s.json' % k))
    print(k, {x: d.get(x) for x in ('graph_construction_s', 'request_phase_s', 'fps_all_inclusive', 'fps_steady', 'identical_to_c_abi', 'error')}, 'lazy', {x: d.get('lazy_super', {}).get(x) for x in ('graph_construction_s', 'request_phase_s', 'fps_all_inclusive', 'fps_steady', 'identical_to_c_abi')})
PY
    grep -v "trace" $out/vs_team/vs_shell_stderr_default.txt | tail -12
    grep "trace" $out/vs_team/vs_shell_stderr_default.txt | head -70
}

r5_side16() {
    # 16x16 blocks side by side (overlap 0) through the row passes: parity, the bench line against the serial kernel, and cfg3 must not have moved
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or analyse_parity or golden or team" 2>&1 | tail -8 | tee $out/r5_side16_tests.txt
    if grep -q "failed\|error" $out/r5_side16_tests.txt; then echo "not green: no timing"; return 1; fi
    {
    timeout 400 python bench.py --config hd16s --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16s (1080p8 16x16 overlap 0) row passes"
    MVX_SPEC=0 timeout 400 python bench.py --config hd16s --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16s serial lean kernel"
    timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 --slots 1 2>&1 | tail -1 | line "cfg3 one batch in flight (must not have moved: 345 ms)"
    timeout 400 python bench.py --config hd16 --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16 (must not have moved: 144 ms)"
    } 2>&1 | tee $out/r5_side16_bench.txt
}

r5_lumaonly() {
    # luma-only searches (chroma = 0) of 16x16 blocks through the row passes (no UV rows): parity, the bench line against the serial kernel, cfg3 must not have moved
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or analyse_parity or golden or team" 2>&1 | tail -8 | tee $out/r5_lumaonly_tests.txt
    if grep -q "failed\|error" $out/r5_lumaonly_tests.txt; then echo "not green: no timing"; return 1; fi
    {
    timeout 400 python bench.py --config hd16l --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16l (1080p8 16x16 overlap 8, chroma=0) row passes"
    MVX_SPEC=0 timeout 400 python bench.py --config hd16l --no-cpu --no-traffic --no-others --steps 2 --warmup 1 2>&1 | tail -1 | line "hd16l serial lean kernel"
    timeout 400 python bench.py --no-cpu --no-traffic --no-others --steps 3 --warmup 1 --slots 1 2>&1 | tail -1 | line "cfg3 one batch in flight (must not have moved: 345 ms)"
    } 2>&1 | tee $out/r5_lumaonly_bench.txt
}

r5_degrain_side() {
    # Degrain of blocks side by side (overlap 0) through the cell kernel instead of the per-sample gather: parity, hd16s and cfg1 bench lines
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_vs_shim.py -x -q -m gpu -k "degrain or golden or full_size or device" 2>&1 | tail -8 | tee $out/r5_degrain_side_tests.txt
    if grep -q "failed\|error" $out/r5_degrain_side_tests.txt; then echo "not green: no timing"; return 1; fi
    {
    timeout 400 python bench.py --config hd16s --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "hd16s (was 135.0 ms/step, Degrain gather 42 ms)"
    timeout 400 python bench.py --config cfg1 --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg1 (r4: 66 700 fps)"
    timeout 400 python bench.py --config cfg4 --no-cpu --no-traffic --no-others --steps 3 --warmup 1 2>&1 | tail -1 | line "cfg4 (unchanged path)"
    } 2>&1 | tee $out/r5_degrain_side_bench.txt
}

s=$1; shift
case "$s" in
  col) r5_col "$@" ;;
  final) r5_final "$@" ;;
  final_light) r5_final_light "$@" ;;
  first) r5_first "$@" ;;
  full) r5_full "$@" ;;
  full2) r5_full2 "$@" ;;
  hd16) r5_hd16 "$@" ;;
  shadow_ab) r5_shadow_ab "$@" ;;
  slots) r5_slots "$@" ;;
  strip32) r5_strip32 "$@" ;;
  team) r5_team "$@" ;;
  team2) r5_team2 "$@" ;;
  team_shapes) r5_team_shapes "$@" ;;
  uni) r5_uni "$@" ;;
  vs_trace) r5_vs_trace "$@" ;;
  side16) r5_side16 "$@" ;;
  lumaonly) r5_lumaonly "$@" ;;
  degrain_side) r5_degrain_side "$@" ;;
  *) echo "unknown session: $s"; exit 2 ;;
esac
