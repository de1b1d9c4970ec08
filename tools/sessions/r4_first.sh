#!/bin/bash
# round 4, first GPU session: the two measurements VERDICT r3 asks for before any kernel work
#   1. per-phase s_memtime cycles of one chain of the lean kernel at 3 / 2 / 1 chains per SIMD (tools/sessions/r4_fastprof.sh)
#   2. tools/micro/ta_pattern.hip incl. patterns 11-15 (rows of a block packed at 32 / 64 / 128-byte pitch)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/sessions/r4_fastprof.sh > /dev/null 2>&1
cat gpurun_out/r4_lean_kernel_phase_cycles.txt
TAG=r4 bash tools/gpu_session.sh micro ta_pattern
