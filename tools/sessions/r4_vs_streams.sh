#!/bin/bash
# round 4: a small pool of per-frame streams in the shell (MVX_VS_FRAME_STREAMS), lazy and default modes, 640 4K16 frames, 32 threads
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_vs_shell_frame_streams.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" VS_ORDER=frame VS_MARKS=1 timeout 600 python tools/vs_4k_run.py 640 32 2>&1 | grep -E "steady state|second half|== batched" | cut -c1-300 >> $O; }
run "warm-up" VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=1
run "lazy, 1 stream" VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=1
run "lazy, 4 streams (verified)" MVX_VS_SUPER_LAZY=1 MVX_VS_FRAME_STREAMS=4
run "lazy, 8 streams" VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=1 MVX_VS_FRAME_STREAMS=8
run "default, 4 streams" VS_NOVERIFY=1 MVX_VS_FRAME_STREAMS=4
cat $O
