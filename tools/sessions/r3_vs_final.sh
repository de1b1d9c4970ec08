#!/bin/bash
# the record run of the VapourSynth shell: 640 frames 4K16 Degrain3, 32 request threads, verified bit for bit; then look-ahead depth 3
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r3_vs_shell_final.txt; : > $O
echo "defaults: look-ahead windows of 128 frames, depth 2, 12 shared search streams, 16 hardware queues" >> $O
VS_MARKS=1 python tools/vs_4k_run.py 640 32 2>&1 | tail -9 >> $O
echo "look-ahead depth 3" >> $O
MVX_VS_LOOKAHEAD_DEPTH=3 VS_NOVERIFY=1 VS_MARKS=1 python tools/vs_4k_run.py 640 32 2>&1 | tail -5 >> $O
echo "16 request threads" >> $O
VS_NOVERIFY=1 VS_MARKS=1 python tools/vs_4k_run.py 640 16 2>&1 | tail -5 >> $O
grep -v "^minihost" $O | cut -c1-400
