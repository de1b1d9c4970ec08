"""developer tool: per-phase cycles of one chain of the lean search kernel, from a library built with -DMVX_FAST_PROF
(python tools/build_variant.py fastprof "MVX_FAST_PROF" mvx_analyse_u16.hip):

    MVX_LIB=tools/variants/fastprof.so python tools/fastprof.py [cfg] [batch]

batch 512 = 3072 chains = three per SIMD (the bench's shape), batch 171 = 1026 chains = one per SIMD.  A stamp does not wait for the vector
loads in flight: a phase is the time the wave spent between two points of its own instruction stream, waiting included."""
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
N = 12
out = (C.c_ulonglong * N)()
assert mv.lib().mvx_debug_fastprof(out) == 0
names = ["loop top + barrier", "group fetch + source block -> LDS", "limits, predictors, lambda", "predictor pass: addresses, loads, SADs", "predictor pass: sums, costs, acceptance",
         "hexagon pass: addresses, loads, SADs", "hexagon pass: sums, costs, acceptance", "square pass", "exhaustive pass (coarser levels)", "rescue", "result"]
nb = int(out[11])
tot = sum(out[:11])
print("one chain, %d blocks (all levels); s_memtime ticks = shader cycles" % nb)
for i, n in enumerate(names):
    print("%-44s %14d cycles  per block %8.1f  %5.1f %%" % (n, out[i], out[i] / max(nb, 1), 100.0 * out[i] / max(tot, 1)))
print("%-44s %14d cycles  per block %8.1f" % ("total", tot, tot / max(nb, 1)))
