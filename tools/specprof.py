"""developer tool: per-phase cycles of one chain of the speculative search kernel, from a library built with -DMVX_SPEC_PROF
(python tools/build_variant.py specprof "MVX_SPEC_PROF" mvx_analyse_spec_u16.hip):

    MVX_LIB=tools/variants/specprof.so python tools/specprof.py [cfg] [batch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import mvtools_amd as mv  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else cfg[6]
p = bench.Pipeline(mv, torch, cfg, batch, torch.device("cuda", 0), 1)
p.step(); p.step(time_search=True)
torch.cuda.synchronize()
print("batch %d: search launch %.1f ms (instrumented build)" % (batch, p.ev[0][0].elapsed_time(p.ev[0][1])))
out = (C.c_ulonglong * 20)()
assert mv.lib().mvx_debug_specprof(out) == 0
names = ["workgroup barrier (waiting for the slowest chain)", "fetch + A1 (predictors, limits, lane-parallel)", "A: row passes (strip / block form)", "A: one block at a time", "A2 (costs, refinement, lane-parallel)",
         "B: verification", "B: live blocks", "results / between groups", "level prologue (interpolation, global motion)"]
tot = sum(out[:9]) + sum(out[11:15]) + out[16] + out[17]
print("one chain: %d groups of up to 32 blocks, %d live blocks; s_memtime ticks = shader cycles" % (out[9], out[10]))
for i, n in enumerate(names):
    print("%-52s %14d cycles  per group %9.1f  %5.1f %%" % (n, out[i], out[i] / max(int(out[9]), 1), 100.0 * out[i] / max(tot, 1)))
for i, n in ((16, "TEAM: waiting for the previous block row"), (17, "TEAM: waiting for the token")):
    print("%-52s %14d cycles  per group %9.1f  %5.1f %%" % (n, out[i], out[i] / max(int(out[9]), 1), 100.0 * out[i] / max(tot, 1)))
print("%-52s %14d cycles" % ("total (this WAVE's groups; a team wave owns every nw-th group)", tot))
np_ = max(int(out[15]), 1)
print("inside the row passes (%d passes): candidates of the next pass %.0f, source strip %.0f, rows (loads + SADs) %.0f, sums + table %.0f, rest %.0f cycles per pass" % (
    np_, out[11] / np_, out[12] / np_, out[13] / np_, out[14] / np_, out[2] / np_))
