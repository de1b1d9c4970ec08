#!/bin/bash
# look-ahead windows of six mv.Analyse instances: do their searches overlap on the device?  (hardware queues x shared stream pool)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r3_vs_search_queues.txt; : > $O
VS_MARKS=1 MVX_VS_TRACE=1 python tools/vs_4k_run.py 640 32 2>&1 | tail -6 > /tmp/first.txt   # writes the clip; verified run; plugin default (8 queues)
echo "shared pool of 8 search streams, GPU_MAX_HW_QUEUES=8 set by the plugin (verified run)" >> $O; cat /tmp/first.txt >> $O
cp gpurun_out/vs_trace_events.txt gpurun_out/vs_trace_events_8queues.txt
for q in 4 16; do
  echo "shared pool of 8 search streams, GPU_MAX_HW_QUEUES=$q" >> $O
  GPU_MAX_HW_QUEUES=$q VS_NOVERIFY=1 VS_MARKS=1 python tools/vs_4k_run.py 640 32 2>&1 | tail -5 >> $O
done
echo "depth 3, plugin default queues" >> $O
MVX_VS_LOOKAHEAD_DEPTH=3 VS_NOVERIFY=1 VS_MARKS=1 python tools/vs_4k_run.py 640 32 2>&1 | tail -5 >> $O
grep -v "^mvtools_vs: thread\|^minihost" $O
