"""developer tool: per-phase cycles of one chain of the LDS-window search kernel (variant built with MVX_WIN_PROF):
MVX_LIB=tools/variants/winprof.so python tools/winprof.py [cfg] [batch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import mvtools_amd as mv
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else cfg[6]
p = bench.Pipeline(mv, torch, cfg, batch, torch.device("cuda", 0), 1)
p.step(); p.step(time_search=True)
torch.cuda.synchronize()
print("search launch %.1f ms (instrumented build)" % p.ev[0][0].elapsed_time(p.ev[0][1]))
out = (C.c_ulonglong * 8)()
assert mv.lib().mvx_debug_winprof(out) == 0
names = ["loop top + barrier", "limits, predictors", "DMA issue", "lambda + DMA wait", "source regs + search passes", "result / store"]
nb = sum(a * b for a, b in p.level_grids())
tot = sum(out[:6])
for i, n in enumerate(names):
    print("%-28s %14d cycles  per block %8.1f  %5.1f %%" % (n, out[i], out[i] / nb, 100.0 * out[i] / tot))
print("%-28s %14d cycles  per block %8.1f" % ("total", tot, tot / nb))
