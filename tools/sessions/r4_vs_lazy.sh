#!/bin/bash
# round 4: the filter shell with lazy super frames (MVX_VS_SUPER_LAZY=1): 640 4K16 frames, 32 request threads, frame order; verified against the C ABI
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_vs_shell_lazy.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" VS_ORDER=frame VS_MARKS=1 timeout 600 python tools/vs_4k_run.py 640 32 2>&1 | grep -E "steady state|second half|progress|thread-seconds|launches=|== batched|shell:" | cut -c1-700 >> $O; }
run "warm-up run (first process on a fresh box pays for the first touch of device memory), default mode" VS_NOVERIFY=1
run "lazy super frames, verified against the C ABI" MVX_VS_SUPER_LAZY=1
run "default mode" VS_NOVERIFY=1
run "lazy super frames" VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=1
run "lazy super frames, look-ahead depth 3" VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=1 MVX_VS_LOOKAHEAD_DEPTH=3
cat $O
