#!/bin/bash
# The GPU sessions of round 6, one function per session.
#     gpurun --timeout 1500 -- 'bash tools/sessions/r6.sh <session>'
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$1', round(d['value'],1), d['unit'], round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step', 'alone', (r.get('launch_alone') or {}).get('ms'), r['kernel'][:60], 'parity', d.get('parity_check',{}).get('identical'))"; }

r6_formats() {
    # r6 first session: the new 4:4:4 / 4:2:2 / Gray cases and the overlap-0 Degrain cases, then the starting point of the round (unchanged library)
    timeout 900 python -m pytest tests/test_gpu_formats.py -q -m gpu 2>&1 | tail -40 | tee $out/r6_formats_tests.txt
    timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "degrain_parity" 2>&1 | tail -8 | tee $out/r6_degrain_side_tests.txt
    timeout 600 python bench.py --no-cpu --no-traffic --no-others --no-vs --steps 5 --warmup 2 2>&1 | tail -1 | tee $out/r6_start_bench.json | line "cfg3 start of r6"
}

r6_ab() {
    # A/B of library builds on cfg3, one batch in flight, with the in-run traffic passes: r6.sh ab <tests -k expression or ""> <lib> [<lib> ...]   (lib "default" = the tree's build)
    k=$1; shift
    for lib in "$@"; do
        if [ "$lib" != default ]; then export MVX_LIB=$PWD/$lib; else unset MVX_LIB; fi
        if [ -n "$k" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$k" 2>&1 | tail -3 | sed "s|^|$lib: |"; fi
        timeout 600 python bench.py --no-cpu --no-others --no-vs --slots 1 --steps 3 --warmup 1 $BENCH_ARGS 2>/dev/null | tail -1 > $out/r6_ab_$(basename $lib .so).json
        python - $out/r6_ab_$(basename $lib .so).json $lib <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
print(sys.argv[2], round(d['value'], 1), 'fps', round(r['avg_launch_ms'], 1), 'ms/launch', round(d['ms_per_step'], 1), 'ms/step traffic', r.get('traffic'), 'parity', d.get('parity_check', {}).get('identical'))
PY
    done 2>&1 | tee $out/r6_ab_${TAG:-last}.txt
}

r6_suite() {
    timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $out/r6_tests_gpu_${TAG:-last}.txt
}

r6_cfg2pmc() {
    # name the bound of the 8-bit search.  Counters of analyse_spec_kernel<1, 8, 2, 8, true> on cfg2 (one launch of 4096 chains), then the phase cycles of one chain
    # (specprof variant), then the same two for hd16 and cfg3 for comparison
    for c in cfg2 hd16; do
        PMC_FILTER=analyse_spec bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" -- python bench.py --config $c --no-cpu --no-parity --no-traffic --steps 1 --warmup 0 --slots 1 > /dev/null 2>&1
        cp $out/pmc_summary.txt $out/r6_${c}_search_counters.txt
    done
    for c in cfg2 hd16 cfg3; do
        echo "== $c"; MVX_LIB=$PWD/tools/variants/specprof.so timeout 300 python tools/specprof.py $c 2>&1 | grep -v amdgpu.ids
    done | tee $out/r6_spec_phase_cycles.txt
}

r6_shadow8() {
    # r6 experiment: three byte-shifted copies of every 8-bit luma plane (every reference load of the 8-bit row passes dword-aligned: 16 instead of 64 texture-path cycles)
    MVX_SHADOW8=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative_kernel_8bit or analyse_parity or degrain_parity or golden or full_size_parity_cfg2" 2>&1 | tail -4
    for c in cfg2 cfg4 hd16; do
        for s8 in 0 1; do
            MVX_SHADOW8=$s8 timeout 600 python bench.py --config $c --no-cpu --no-traffic --slots 1 --steps 3 --warmup 1 2>/dev/null | tail -1 | line "$c shadow8=$s8"
        done
    done 2>&1 | tee $out/r6_shadow8_ab.txt
}

r6_stats() {
    # strip / block windows of stage 2 per level (specstats variant), phase cycles of one chain (specprof variant)
    MVX_LIB=$PWD/tools/variants/specstats.so timeout 300 python tools/specstats.py cfg3 341 2>&1 | grep -v amdgpu.ids | tee $out/r6_spec_window_stats.txt
    MVX_LIB=$PWD/tools/variants/specprof.so timeout 300 python tools/specprof.py cfg3 2>&1 | grep -v amdgpu.ids | tee $out/r6_spec_phase_cycles_cfg3.txt
}

r6_k34() {
    # chains per SIMD of the 8-bit row-pass builds: 2 (256 registers, the default), 3 (168), 4 (128)
    for c in cfg2 cfg4 hd16; do
        for k in 0 3 4; do
            MVX_FAST_K=$k timeout 600 python bench.py --config $c --no-cpu --no-traffic --slots 1 --steps 3 --warmup 1 2>/dev/null | tail -1 | line "$c fast_k=$k"
        done
    done 2>&1 | tee $out/r6_8bit_chains_per_simd.txt
}

r6_configs() {
    # every bench configuration, one batch in flight, 3 timed steps each (TAG names the state)
    for c in cfg3 cfg2 cfg4 hd16 hd16s cfg1 cfg5; do
        timeout 600 python bench.py --config $c --no-cpu --no-traffic --no-others --no-vs --slots 1 --steps 3 --warmup 1 2>/dev/null | tail -1 | line "$c"
    done 2>&1 | tee $out/r6_configs_${TAG:-last}.txt
}

r6_final() {
    # r6 final state: the whole GPU suite, smoke(), the driver's bench command (20 steps like the driver), rocprofv3 kernel stats of the default command, per-kernel HBM traffic,
    # SQ / TCP / TCC counters of the search kernel
    timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $out/r6_tests_gpu_final.txt
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/r6_smoke.txt
    t0=$(date +%s)
    timeout 1500 python bench.py --steps 20 --warmup 5 > $out/r6_bench_default.json 2> $out/r6_bench_default.err || tail -5 $out/r6_bench_default.err
    echo "bench.py --steps 20 --warmup 5 wall: $(( $(date +%s) - t0 )) s"
    python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_bench_default.json') if l.startswith('{')][-1])
r=d['roofline']
print('default', round(d['value'],1), 'fps', round(r['avg_launch_ms'],1), 'ms/launch', round(d['ms_per_step'],1), 'ms/step frac', round(r['frac'],4), 'alone', r.get('launch_alone',{}).get('avg_launch_ms'), 'traffic', r['traffic'], 'parity', d['parity_check']['identical'], 'cpu', d['cpu_baseline']['value'], 'one batch', d.get('one_batch_in_flight',{}).get('value'))
print('others', {k:(round(v.get('fps',0),1) if isinstance(v,dict) else v) for k,v in d.get('other_configs',{}).items()})
print('vs', {k:d.get('vs_shell',{}).get(k) for k in ('fps_all_inclusive','fps_steady','identical_to_c_abi','graph_construction_s','request_phase_s')}, (d.get('vs_shell',{}).get('lazy_super') or {}).get('fps_all_inclusive'))
PY
    TAG=r6 BENCH_ARGS="--no-others --no-vs" bash tools/gpu_session.sh stats traffic > $out/r6_profile_steps.log 2>&1
    PMC_FILTER=analyse_spec bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python bench.py --no-cpu --no-parity --no-traffic --no-others --no-vs --steps 1 --warmup 0 --slots 1 > /dev/null 2>&1
    cp $out/pmc_summary.txt $out/r6_search_final_counters.txt
    head -14 $out/r6_kernel_stats.csv | cut -c1-200
}

r6_vs_sweep() {
    # the filter shell on 640 4K16 frames, default mode: download streams x request threads x look-ahead window (the clip file is written once; only the first run is verified against the C ABI)
    timeout 600 python tools/vs_4k_run.py 640 48 2>&1 | grep -v amdgpu.ids | tail -6
    for v in "2 48 128 2" "3 48 128 2" "4 48 128 2" "2 64 128 2" "3 64 128 2" "2 96 128 2" "2 48 64 2" "2 48 64 3" "2 48 96 2" "2 48 192 2" "3 64 64 3" "2 48 128 3"; do
        set -- $v
        echo "== dl_streams=$1 threads=$2 lookahead=$3 depth=$4"
        VS_NOVERIFY=1 VS_MARKS=1 MVX_VS_DL_STREAMS=$1 MVX_VS_LOOKAHEAD=$3 MVX_VS_LOOKAHEAD_DEPTH=$4 timeout 300 python tools/vs_4k_run.py 640 $2 2>&1 | grep -E "minihost: (graph built|output clip)|steady state|progress|shell:" | cut -c1-260
    done 2>&1 | tee $out/r6_vs_sweep.txt
}

r6_vs_sweep2() {
    # the same, three repeats per setting (run-to-run noise of the request phase is +-0.2 s): download streams x request threads
    timeout 600 python tools/vs_4k_run.py 640 48 2>&1 | grep -v amdgpu.ids | tail -3
    for rep in 1 2 3; do
        for v in "2 48" "3 48" "4 48" "2 64" "3 64" "4 64" "2 96" "3 96" "4 96"; do
            set -- $v
            r=$(VS_NOVERIFY=1 MVX_VS_DL_STREAMS=$1 timeout 300 python tools/vs_4k_run.py 640 $2 2>&1 | grep -E "minihost: output clip" | sed 's/.*order) //')
            echo "rep $rep dl_streams=$1 threads=$2 request phase $r"
        done
    done 2>&1 | tee $out/r6_vs_sweep2.txt
}

r6_degrain() {
    # Degrain cell kernel: the plan records of a workgroup's tile in LDS (default build) against every thread reading its own (dgnotile), and the chroma kernel of
    # Degrain3 at seven instead of eight waves per SIMD (dgw4_7): parity subset, then per-kernel durations with ONE batch in flight (nothing overlaps the Degrain kernels)
    for lib in "$@"; do
        if [ "$lib" != default ]; then export MVX_LIB=$PWD/$lib; else unset MVX_LIB; fi
        timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats.py -x -q -m gpu -k "degrain or golden or full_size" 2>&1 | tail -2 | sed "s|^|$lib: |"
        for c in cfg3 cfg2 hd16 cfg5; do
            (cd /tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $OLDPWD/bench.py --config $c --no-cpu --no-parity --no-traffic --no-others --no-vs --slots 1 --steps 3 --warmup 1 > /tmp/kt.log 2>&1)
            f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1)
            python - $f "$lib" $c <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'degrain' in r['Name'] or 'usable' in r['Name']:
        print(sys.argv[2], sys.argv[3], r['Name'][:64].ljust(64), 'calls', r['Calls'], 'avg %.3f ms' % (float(r['AverageNs']) / 1e6))
PY
            grep '^{' /tmp/kt.log | tail -1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib $c', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms/step')"
        done
    done 2>&1 | tee $out/r6_degrain_tile_ab.txt
}

r6_vs_threads() {
    # request threads of the host beyond 96 (a VapourSynth core starts one worker per logical CPU), two repeats each
    timeout 600 python tools/vs_4k_run.py 640 96 2>&1 | grep -v amdgpu.ids | tail -3
    for rep in 1 2; do
        for t in 96 128 192 256; do
            r=$(VS_NOVERIFY=1 VS_MARKS=1 timeout 300 python tools/vs_4k_run.py 640 $t 2>&1 | grep -E "minihost: (graph built|output clip)|progress" | sed 's/.*order) //' | tr '\n' ' ')
            echo "rep $rep threads=$t request phase $r"
        done
    done 2>&1 | tee $out/r6_vs_threads.txt
}

r6_halfslots() {
    # 8-bit configurations (equally long chains: a full launch frees its wave slots all at once, Super / Degrain of the neighbouring batch have nothing to run under):
    # launches of HALF the wave slots (batch 512 = 1024 chains = one wave per SIMD, one wave per chain: MVX_TEAM=0) from three or four batches in flight -- two searches
    # resident, the other batches' Super / Degrain kernels in the registers a finished launch frees
    for c in cfg2 cfg4 hd16; do
        timeout 600 python bench.py --config $c --no-cpu --no-traffic --no-others --no-vs --steps 8 --warmup 2 2>/dev/null | grep '^{' | tail -1 | line "$c default (2 x 2048)"
        for v in "2 512" "3 512" "4 512" "6 512" "3 1024"; do  # (MVX_FAST_K=2: the two-waves-per-SIMD build -- the one-wave build takes more than 256 registers, two launches could not share a SIMD)
            set -- $v
            MVX_TEAM=0 MVX_FAST_K=2 timeout 600 python bench.py --config $c --slots $1 --batch $2 --no-cpu --no-traffic --no-others --no-vs --steps $(( 16 * 2048 / $2 / $1 )) --warmup $1 2>/dev/null | grep '^{' | tail -1 | line "$c slots $1 batch $2 team 0"
        done
        MVX_FAST_K=2 timeout 600 python bench.py --config $c --slots 4 --batch 512 --no-cpu --no-traffic --no-others --no-vs --steps 16 --warmup 4 2>/dev/null | grep '^{' | tail -1 | line "$c slots 4 batch 512 library team choice"
    done 2>&1 | tee $out/r6_half_occupancy_launches.txt
}

r6_vs_threads_stats() {
    # where the shell's time goes with 96 / 128 / 192 request threads (MVX_VS_STATS thread-seconds)
    timeout 600 python tools/vs_4k_run.py 640 96 2>&1 | grep -v amdgpu.ids | tail -3
    for t in 96 128 192; do
        echo "== threads=$t"
        VS_NOVERIFY=1 VS_MARKS=1 timeout 300 python tools/vs_4k_run.py 640 $t 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-1500
    done 2>&1 | tee $out/r6_vs_threads_stats.txt
}

r6_vs_gate() {
    # the admission gate of the consuming filters (MVX_VS_MAX_INFLIGHT, default 96): request threads 48 .. 256 with the gate, the limit itself at 256 threads, and the gate off
    timeout 900 python -m pytest tests/test_vs_shim.py -x -q -m gpu 2>&1 | tail -2
    timeout 600 python tools/vs_4k_run.py 640 256 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400
    for v in "96 48" "96 96" "96 128" "96 192" "96 256" "64 256" "80 256" "112 256" "128 256" "0 96" "0 128" "96 256" "96 96"; do
        set -- $v
        r=$(VS_NOVERIFY=1 VS_MARKS=1 MVX_VS_MAX_INFLIGHT=$1 timeout 300 python tools/vs_4k_run.py 640 $2 2>&1 | grep -E "minihost: output clip|progress" | sed 's/.*order) //' | tr '\n' ' ')
        echo "max_inflight=$1 threads=$2 request phase $r"
    done 2>&1 | tee $out/r6_vs_gate.txt
}

r6_cfg5k3() {
    # cfg5 (8K16, 32x32 blocks, serial lean kernel): three chains per SIMD need a batch of 252 frames, which only fits without the shifted luma copies (MVX_SHADOW=0: 0.52 instead of 1.05 GB per super frame)
    for v in "1 168" "0 168" "0 252"; do
        set -- $v
        MVX_SHADOW=$1 timeout 900 python bench.py --config cfg5 --batch $2 --no-cpu --no-traffic --no-others --no-vs --slots 1 --steps 2 --warmup 1 2>&1 | grep '^{' | tail -1 | line "cfg5 shadow=$1 batch $2"
    done 2>&1 | tee $out/r6_cfg5_three_per_simd.txt
}

r6_vs_gate2() {
    # the gate with one condition variable per waiter (a delivered frame wakes one request, not all): request threads 64 .. 256, two repeats
    timeout 600 python tools/vs_4k_run.py 640 256 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400
    for rep in 1 2; do
        for t in 64 96 128 192 256; do
            r=$(VS_NOVERIFY=1 VS_MARKS=1 timeout 300 python tools/vs_4k_run.py 640 $t 2>&1 | grep -E "minihost: output clip|progress" | sed 's/.*order) //' | tr '\n' ' ')
            echo "rep $rep max_inflight=96 threads=$t request phase $r"
        done
    done 2>&1 | tee $out/r6_vs_gate2.txt
}

r6_vs_build() {
    # helper threads that upload a look-ahead window's source frames (MVX_VS_BUILD_THREADS, default 4): the bench's own shell leg (graph construction + request phase, default mode; one request thread per logical CPU)
    for rep in 1 2; do
        for b in 4 8 16; do
            MVX_VS_BUILD_THREADS=$b timeout 600 python bench.py --vs-shell-leg 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d.get('lazy_super',{})
print('rep $rep build_threads=$b threads', d.get('threads'), 'graph', d.get('graph_construction_s'), 'requests', d.get('request_phase_s'), 'all inclusive', round(d.get('fps_all_inclusive',0),1), 'fps steady', round(d.get('fps_steady',0),1), 'identical', d.get('identical_to_c_abi'), '| lazy', round(l.get('fps_all_inclusive',0),1), l.get('identical_to_c_abi'))"
        done
    done 2>&1 | tee $out/r6_vs_build_threads.txt
}

r6_vs_gate_limit() {
    # the gate's limit in both modes of mv.Super (default: frames to the host, PCIe-bound; lazy: pixels stay on the device), 256 request threads; 48 threads for reference
    timeout 600 python tools/vs_4k_run.py 640 256 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
    for lazy in 0 1; do
        for v in "32 256" "48 256" "64 256" "96 256" "128 256" "0 48"; do
            set -- $v
            r=$(VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=$lazy MVX_VS_MAX_INFLIGHT=$1 timeout 300 python tools/vs_4k_run.py 640 $2 2>&1 | grep -E "minihost: output clip" | sed 's/.*order) //' | tr '\n' ' ')
            echo "lazy=$lazy max_inflight=$1 threads=$2 request phase $r"
        done
    done 2>&1 | tee $out/r6_vs_gate_limit.txt
}

r6_vs_pad() {
    # (experiment, removed from the tree: the switch no longer exists) the common source-frame range of a look-ahead window (MVX_VS_LOOKAHEAD_PAD, default 1) against every instance its own range (0): shell tests, then the bench's shell leg, three repeats
    timeout 900 python -m pytest tests/test_vs_shim.py -x -q -m gpu 2>&1 | tail -2
    for rep in 1 2 3; do
        for pad in 1 0; do
            MVX_VS_LOOKAHEAD_PAD=$pad timeout 600 python bench.py --vs-shell-leg 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d.get('lazy_super',{})
print('rep $rep pad=$pad threads', d.get('threads'), 'graph', d.get('graph_construction_s'), 'requests', d.get('request_phase_s'), 'all inclusive', round(d.get('fps_all_inclusive',0),1), 'fps steady', round(d.get('fps_steady',0),1), 'identical', d.get('identical_to_c_abi'), '| lazy graph', l.get('graph_construction_s'), 'requests', l.get('request_phase_s'), 'all inclusive', round(l.get('fps_all_inclusive',0),1), l.get('identical_to_c_abi'))"
        done
    done 2>&1 | tee $out/r6_vs_lookahead_pad.txt
}

r6_vs_adata() {
    # (experiment, removed from the tree: the switch no longer exists) consumers take the analysis data from this plugin's own mv.Analyse instance instead of reading frame 0 at creation (MVX_VS_ADATA_FROM_INSTANCE, default 1) against reading it (0):
    # shell tests, then the bench's shell leg, three repeats
    timeout 900 python -m pytest tests/test_vs_shim.py -x -q -m gpu 2>&1 | tail -2
    for rep in 1 2 3; do
        for on in 1 0; do
            MVX_VS_ADATA_FROM_INSTANCE=$on timeout 600 python bench.py --vs-shell-leg 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d.get('lazy_super',{})
print('rep $rep from_instance=$on threads', d.get('threads'), 'graph', d.get('graph_construction_s'), 'requests', d.get('request_phase_s'), 'all inclusive', round(d.get('fps_all_inclusive',0),1), 'fps steady', round(d.get('fps_steady',0),1), 'identical', d.get('identical_to_c_abi'), '| lazy graph', l.get('graph_construction_s'), 'requests', l.get('request_phase_s'), 'all inclusive', round(l.get('fps_all_inclusive',0),1), l.get('identical_to_c_abi'))"
        done
    done 2>&1 | tee $out/r6_vs_adata_from_instance.txt
}

r6_vs_cache() {
    # the mini host's per-node frame cache in the bench's shell leg: unbounded (every one of the 640 131 MB super frames stays in host memory: the default so far) against 160 / 320 frames per node
    # (a real core's caches are bounded; a recycled frame buffer costs no page faults)
    for rep in 1 2; do
        for c in "" 320 160; do
            MVX_VS_BENCH_CACHE=$c timeout 600 python bench.py --vs-shell-leg 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d.get('lazy_super',{})
print('rep $rep cache=${c:-all} threads', d.get('threads'), 'graph', d.get('graph_construction_s'), 'requests', d.get('request_phase_s'), 'all inclusive', round(d.get('fps_all_inclusive',0),1), 'fps steady', round(d.get('fps_steady',0),1), 'identical', d.get('identical_to_c_abi'), '| lazy graph', l.get('graph_construction_s'), 'requests', l.get('request_phase_s'), 'all inclusive', round(l.get('fps_all_inclusive',0),1), l.get('identical_to_c_abi'))"
        done
    done 2>&1 | tee $out/r6_vs_host_cache.txt
}

r6_vs_after() {
    # why the shell leg is ~5 % slower inside the full bench command than on its own: the leg alone, right after another bench child run, the same with a longer pause before the hosts start, after a hot cfg3 run
    leg() { python bench.py --vs-shell-leg 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d.get('lazy_super',{})
print('$1: graph', d.get('graph_construction_s'), 'requests', d.get('request_phase_s'), 'all inclusive', round(d.get('fps_all_inclusive',0),1), '| lazy', round(l.get('fps_all_inclusive',0),1))"; }
    {
    leg "alone (fresh box)"
    timeout 600 python bench.py --config hd16 --no-cpu --no-traffic --no-others --no-vs --steps 8 --warmup 2 > /dev/null 2>&1; leg "right after a hd16 child run"
    timeout 600 python bench.py --config hd16 --no-cpu --no-traffic --no-others --no-vs --steps 8 --warmup 2 > /dev/null 2>&1; MVX_VS_SETTLE_S=20 leg "after a hd16 child run, 20 s pause before each host"
    timeout 600 python bench.py --no-cpu --no-traffic --no-others --no-vs --steps 40 --warmup 5 > /dev/null 2>&1; leg "right after 45 cfg3 steps (hot GPU)"
    leg "alone again"
    } 2>&1 | tee $out/r6_vs_leg_after_other_runs.txt
}

r6_vs_order() {
    # (the MVX_BENCH_VS_FIRST switch was removed from bench.py after this run) the shell leg inside the full bench command: after the other configurations' child runs (the default order) against before them
    for v in "" 1; do
        MVX_BENCH_VS_FIRST=$v timeout 1200 python bench.py --steps 5 --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json
d=json.loads(sys.stdin.read()); v=d.get('vs_shell',{}); l=v.get('lazy_super',{})
print('vs first=${v:-0}:', round(d['value'],1), 'fps | shell graph', v.get('graph_construction_s'), 'requests', v.get('request_phase_s'), 'all inclusive', round(v.get('fps_all_inclusive',0),1), '| lazy', round(l.get('fps_all_inclusive',0),1), '| others', {k:round(x.get('fps',0),1) for k,x in d.get('other_configs',{}).items()})"
    done 2>&1 | tee $out/r6_vs_leg_order_in_bench.txt
}

r6_vs_trim() {
    # (the malloc_trim / MVX_BENCH_NO_TRIM code was removed from bench.py after this run) the shell leg inside the full bench command with the parent's freed host memory returned to the kernel first (malloc_trim) against without
    for v in "" 1; do
        MVX_BENCH_NO_TRIM=$v timeout 1200 python bench.py --steps 5 --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json
d=json.loads(sys.stdin.read()); v=d.get('vs_shell',{}); l=v.get('lazy_super',{})
print('no trim=${v:-0}:', round(d['value'],1), 'fps | shell graph', v.get('graph_construction_s'), 'requests', v.get('request_phase_s'), 'all inclusive', round(v.get('fps_all_inclusive',0),1), '| lazy', round(l.get('fps_all_inclusive',0),1))"
        grep -E "MemFree|MemAvailable|^Cached" /proc/meminfo | tr '\n' ' '; echo; numactl --hardware 2>/dev/null | grep -E "available|free" | tr '\n' ' '; echo
    done 2>&1 | tee $out/r6_vs_leg_parent_trim.txt
}

r6_vs_ctx() {
    # does another process that merely HOLDS a HIP context (as bench.py's parent does while its child runs work) slow the shell leg down?
    leg() { python bench.py --vs-shell-leg 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d.get('lazy_super',{})
print('$1: graph', d.get('graph_construction_s'), 'requests', d.get('request_phase_s'), 'all inclusive', round(d.get('fps_all_inclusive',0),1), '| lazy', round(l.get('fps_all_inclusive',0),1))"; }
    {
    leg "alone"
    python -c "import torch,time; x=torch.zeros(1,device='cuda'); s=[torch.cuda.Stream() for _ in range(3)]; [torch.zeros(1,device='cuda') for _ in s]; torch.cuda.synchronize(); time.sleep(400)" &
    idle=$!
    sleep 8
    leg "beside an idle process that holds a HIP context and three streams"
    leg "the same again"
    kill $idle
    sleep 2
    leg "alone again"
    } 2>&1 | tee $out/r6_vs_leg_beside_idle_context.txt
}

"r6_$1" "${@:2}"
