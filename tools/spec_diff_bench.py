"""developer tool: where the GPU search and the oracle differ on frames of bench.py's own clip (block positions per level).
    python tools/spec_diff_bench.py <cfg> <frame> <delta> <isb> [total_frames]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vapoursynth-mvtools_amd")]
import numpy as np
import torch
import bench
import mvoracle as mo
import mvtools_amd as mv
import pipeline as pl
from collections import Counter

cfgname, n, d, isb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
(w, h, bits, tr, akw, skw, B, label) = bench.CONFIGS[cfgname]
total = int(sys.argv[5]) if len(sys.argv) > 5 else B + 2 * tr
nref = n + d if isb else n - d
lo = min(n, nref)
dev = torch.device("cuda", 0)
fr = bench.synth_clip_device(torch, w, h, bits, abs(nref - n) + 1, 1000, dev, first_frame=lo, total_frames=total)
cur, ref = fr[n - lo], fr[nref - lo]
gsup, osup = mv.Super(w, h, bits, **skw), mo.Super(w, h, bits, **skw)
gsf = gsup.build([cur, ref])
torch.cuda.synchronize()
host = [bench._frame_to_numpy(mv, f, w, h, bits) for f in (cur, ref)]
osf = [osup.frame(f) for f in host]
oan, gan = mo.Analyse(osup, isb=isb, delta=d, **akw), mv.Analyse(gsup, isb=isb, delta=d, **akw)
ob = oan.frame(osf[0], osf[1])
import ctypes as C
at = [int(v) for v in os.environ.get("SPECDBG_AT", "").split(",") if v]
if at and hasattr(mv.lib(), "mvx_debug_specdbg_at"):
    assert mv.lib().mvx_debug_specdbg_at(at[0], at[1], at[2] & ~63) == 0
gb = gan.run([(gsf[0], gsf[1])])[0].cpu().numpy()
info = (__import__("ctypes").c_int * 5)()
mv.lib().mvx_debug_last_launch(info)
print("launch", list(info), "equal", np.array_equal(ob, gb))
for lv in range(oan.ad.nLvCount):
    ox, oy, osad = pl.blob_vectors(ob, oan.ad, lv)
    gx, gy, gsad = pl.blob_vectors(gb, oan.ad, lv)
    bad = np.argwhere((ox != gx) | (oy != gy) | (osad != gsad))
    print("level", lv, "blocks", ox.shape, "differ", len(bad), "first", bad[:10].tolist())
    if len(bad):
        print("   rows:", sorted(Counter(int(b[0]) for b in bad).items())[:12], " columns mod 32:", sorted(Counter(int(b[1] % 32) for b in bad).items())[:32])
        for b in bad[:6]:
            r, c = int(b[0]), int(b[1])
            print("   block", (r, c), "oracle", int(ox[r, c]), int(oy[r, c]), int(osad[r, c]), "gpu", int(gx[r, c]), int(gy[r, c]), int(gsad[r, c]),
                  " up (oracle)", (int(ox[r - 1, c]), int(oy[r - 1, c])) if r else None, " left/right", (int(ox[r, c - 1]), int(oy[r, c - 1])) if c else None, (int(ox[r, c + 1]), int(oy[r, c + 1])) if c + 1 < ox.shape[1] else None)

if at and hasattr(mv.lib(), "mvx_debug_specdbg"):
    buf = (C.c_int * (64 * 24))()
    torch.cuda.synchronize()
    assert mv.lib().mvx_debug_specdbg(buf) == 0
    a = np.array(buf[:]).reshape(64, 24)
    up = lambda v: (int(np.int16(v & 0xffff)), int(v >> 16))
    lv, r, c = at
    ox, oy, osad = pl.blob_vectors(ob, oan.ad, lv)
    for col in range(max(c - 4, c & ~63), min(c + 4, (c & ~63) + 64)):
        o = a[col & 63]
        print("col", col, "oracle", (int(ox[r, col]), int(oy[r, col]), int(osad[r, col])) if col < ox.shape[1] else None, "| U", up(o[0]), "Ah", up(o[1]), "G", up(o[2]), "H", up(o[3]), "W", up(o[4]), "pBest", o[5],
              "tots U/Ah/Z/G/H", o[6:11].tolist(), "lam", o[11], "| spec result", (o[12], o[13], o[14]), "flag/ok/staged/live", bin(o[15]), "prev@A2", up(o[16]), "best@A2", o[17],
              "| B path", {0: "-", 1: "run", 2: "redo", 3: "live"}.get(int(o[18])), "L", up(o[19]), "M", up(o[20]), "fL,fM", (o[21] & 255, o[21] >> 8), "best", o[22], "W/result", up(o[23]))
