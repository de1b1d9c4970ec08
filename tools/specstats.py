"""developer tool: how many blocks of the speculative search kernel (mvx_analyse_spec.h) verify, per level, from a library built with
-DMVX_SPEC_STATS (python tools/build_variant.py specstats "MVX_SPEC_STATS" mvx_analyse_spec_u16.hip):

    MVX_LIB=tools/variants/specstats.so python tools/specstats.py [cfg] [batch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import mvtools_amd as mv  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = bench.Pipeline(mv, torch, cfg, batch, torch.device("cuda", 0), 1)
out = (C.c_ulonglong * (24 * 8))()  # MVX_MAX_LEVELS x 8
p.step()
torch.cuda.synchronize()
assert mv.lib().mvx_debug_specstats(out, 1) == 0
p.step(time_search=True)
torch.cuda.synchronize()
assert mv.lib().mvx_debug_specstats(out, 1) == 0
print("batch %d: search launch %.1f ms (counting build)" % (batch, p.ev[0][0].elapsed_time(p.ev[0][1])))
print("level: blocks in speculated rows, searched live (share), of them flag clear (hexagon won / rescue), accepted after redoing the predictor phase with the true left / median")
for lv in range(16):
    b, live, flag, resc, ws, wb, dev, lim = (int(out[lv * 8 + i]) for i in range(8))
    if b:
        print("%5d: %12d %12d (%5.2f %%) %12d %10d" % (lv, b, live, 100.0 * live / b, flag, resc))
        if ws + wb:
            print("       stage-2 windows (16x16 row passes): strip form %d (%.1f %%), block form %d -- %.2f blocks per block-form window whose centre differs from the window's first block, %d block-form windows for the limits alone" % (
                ws, 100.0 * ws / (ws + wb), wb, dev / max(wb, 1), lim))
