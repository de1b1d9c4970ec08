"""developer tool: where the GPU search and the oracle differ on a small clip (block positions per level)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vapoursynth-mvtools_amd")]
import numpy as np
import mvoracle as mo, mvtools_amd as mv, pipeline as pl
w, h = int(sys.argv[1]), int(sys.argv[2])
frames = pl.moving_clip(w, h, 16, 3, seed=17, noise=int(sys.argv[3]) if len(sys.argv) > 3 else 3, motion=(5, -2))
osup, gsup = mo.Super(w, h, 16), mv.Super(w, h, 16)
osf = [osup.frame(f) for f in frames]
gsf = gsup.build([mv.frame_to_device(f) for f in frames])
akw = dict(blksize=16, overlap=8)
oan, gan = mo.Analyse(osup, isb=1, **akw), mv.Analyse(gsup, isb=1, **akw)
ob = oan.frame(osf[1], osf[2]); gb = gan.run([(gsf[1], gsf[2])])[0].cpu().numpy()
print("equal", np.array_equal(ob, gb))
for lv in range(oan.ad.nLvCount):
    ox, oy, osad = pl.blob_vectors(ob, oan.ad, lv); gx, gy, gsad = pl.blob_vectors(gb, oan.ad, lv)
    bad = np.argwhere((ox != gx) | (oy != gy) | (osad != gsad))
    print("level", lv, "blocks", ox.shape, "differ", len(bad), "first", bad[:12].tolist())
    if len(bad):
        from collections import Counter
        print("   columns mod 32:", sorted(Counter((b[1] % 32) for b in bad).items())[:40])
        b = bad[0]; print("   oracle", ox[b[0], b[1]], oy[b[0], b[1]], osad[b[0], b[1]], "gpu", gx[b[0], b[1]], gy[b[0], b[1]], gsad[b[0], b[1]])
lv = 1
ox, oy, osad = pl.blob_vectors(ob, oan.ad, lv); gx, gy, gsad = pl.blob_vectors(gb, oan.ad, lv)
for r in range(ox.shape[0]):
    print("row", r, "oracle", [(int(ox[r, c]), int(oy[r, c]), int(osad[r, c])) for c in range(50, ox.shape[1])])
    print("row", r, "gpu   ", [(int(gx[r, c]), int(gy[r, c]), int(gsad[r, c])) for c in range(50, ox.shape[1])])

import ctypes as C
if hasattr(mv.lib(), "mvx_debug_specdbg"):
    buf = (C.c_int * (4 * 64 * 12))()
    # (the dump holds whichever chain wrote last: run the single forward job again alone)
    gan.run([(gsf[1], gsf[2])]); import torch; torch.cuda.synchronize()
    mv.lib().mvx_debug_specdbg(buf)
    a = np.array(buf[:]).reshape(4, 64, 12)
    def up(v): return (int(np.int16(v & 0xffff)), int(v >> 16))
    for r in range(2):
        for c in range(50, 61):
            o = a[r, c]
            print("row", r + 1, "col", c, "U", up(o[0]), "Ah", up(o[1]), "G", up(o[2]), "H", up(o[3]), "W", up(o[4]), "best", o[5], "tots U/Ah/Z/G/H", o[6:11].tolist(), "lam", o[11])
