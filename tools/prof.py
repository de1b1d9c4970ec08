"""developer tool: per-phase cycle counters of one chain of the search kernel (library built with MVX_PROFILE=1).
usage: MVX_ABLATE=$((chain<<8)) python tools/prof.py [cfg] [batch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import mvtools_amd as mv
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else cfg[6]
p = bench.Pipeline(mv, torch, cfg, batch, "cuda:0", 1)
p.step(); p.step()
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
assert mv.lib().mvx_debug_prof(out) == 0
names = ["prologue", "search", "epilogue", "blocks", "eval(load+sad)", "passes with window miss", "reduce+argmin", "7", "passes", "kernel total",
         "sm: setup", "sm: round", "sm: post", "pro: fetch+stage", "pro: prefetch", "pro: window"]
nb = max(1, out[3])
for i, n in enumerate(names):
    print("%2d %-26s %14d  per block %9.1f" % (i, n, out[i], out[i] / nb))
