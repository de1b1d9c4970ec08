#!/bin/bash
# round 4: the filter shell on 640 4K16 frames, 32 request threads, frame order: default / look-ahead autosizing off / the serial search kernel / depth 3
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_vs_shell_ab.txt; : > $O
run() { echo "== $1" >> $O; shift; env "$@" VS_ORDER=frame VS_MARKS=1 timeout 600 python tools/vs_4k_run.py 640 32 2>&1 | grep -E "steady state|second half|progress|thread-seconds|launches=|== batched" | cut -c1-700 >> $O; }
run "default (verifies against the C ABI)"
run "look-ahead autosizing off" VS_NOVERIFY=1 MVX_VS_LOOKAHEAD_AUTOSIZE=0
run "serial search kernel" VS_NOVERIFY=1 MVX_SPEC=0
run "look-ahead depth 3" VS_NOVERIFY=1 MVX_VS_LOOKAHEAD_DEPTH=3
run "default again" VS_NOVERIFY=1
cat $O
