#!/bin/bash
# round 4: the filter shell with post-sync downloads on the download stream: default and lazy modes (640 4K16 frames, 32 threads, frame order)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_vs_shell_download_stream.txt; : > $O
run() { echo "== $1" >> $O; shift; T=$1; shift; env "$@" VS_ORDER=frame VS_MARKS=1 timeout 600 python tools/vs_4k_run.py 640 $T 2>&1 | grep -E "steady state|second half|progress|thread-seconds|== batched" | cut -c1-600 >> $O; }
run "warm-up (first process on the box)" 32 VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=1
run "lazy super frames, verified" 32 MVX_VS_SUPER_LAZY=1
run "default mode" 32 VS_NOVERIFY=1
run "lazy super frames, 48 request threads" 48 VS_NOVERIFY=1 MVX_VS_SUPER_LAZY=1
cat $O
