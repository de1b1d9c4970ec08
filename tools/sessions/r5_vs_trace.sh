#!/bin/bash
# r5: where the shell's graph construction (3 s of a 5.9 s run) goes: window trace + thread-second statistics, 384 frames; team form off for comparison
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out/vs_team $out/vs_noteam
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
