#!/usr/bin/env python3
"""Developer tool (CPU only): how the default motion search behaves on a clip, counted by the oracle built with -DMVO_STATS (a private
copy of the library under /tmp; the test library carries no counters).  Per level of the hierarchy: which of the seven predictor
candidates a block's predictor phase ends on, whether that vector equals the hierarchical predictor / the median / the left neighbour,
how many DIFFERENT vectors the seven candidates are, how often a point of the first hexagon wins, how often the bad-block rescue runs.
These are the probabilities a kernel that wants to overlap the passes of a block by speculation needs.

    python tools/search_stats.py [width height bits frames]        (default: the bench clip, 3840 2160 16, 3 frames)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import mvoracle as mo  # noqa: E402
import pipeline as pl  # noqa: E402

so = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libmvoracle_stats.so")
subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-mavx2", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", "-DMVO_STATS", "-shared", "-o", so]
                      + [os.path.join(ROOT, "oracle", f) for f in ("mvo_super.c", "mvo_analyse.c", "mvo_degrain.c", "mvo_blockfps.c")] + ["-lm"])
mo.build = lambda force=False: so  # this process counts

w, h, bits, n = (int(v) for v in (sys.argv[1:5] + ["3840", "2160", "16", "3"][len(sys.argv) - 1:]))
noise = int(os.environ.get("NOISE", "2"))
bs = int(os.environ.get("BLK", "16"))
frames = pl.moving_clip(w, h, bits, n, seed=3, noise=noise)
sup = mo.Super(w, h, bits)
sf = [sup.frame(f) for f in frames]
L = mo.lib()
L.mvo_stats_reset()
jobs = 0
for isb in (1, 0):
    an = mo.Analyse(sup, isb=isb, delta=1, blksize=bs, overlap=bs // 2)
    for k in range(n):
        r = k + 1 if isb else k - 1
        if 0 <= r < n:
            an.frame(sf[k], sf[r]); jobs += 1
NF = L.mvo_stats_fields()
levels = an.ad.nLvCount
buf = (C.c_longlong * (NF * 16))()
L.mvo_stats_get(buf, 16)
st = np.array(buf[:], dtype=np.int64).reshape(16, NF)
names = ["zero", "global", "hierarchical", "median", "left", "up", "ahead"]
print("%dx%d %d-bit, %d searches (blksize 16, overlap 8, pel 2, default search), per level (0 = finest):" % (w, h, bits, jobs))
for lv in range(levels):
    s = st[lv]
    nb = int(s[0])
    if not nb:
        continue
    pct = lambda v: "%5.1f %%" % (100.0 * v / nb)
    print("level %d: %d blocks" % (lv, nb))
    print("   predictor phase ends on (first candidate with the winning vector): " + ", ".join("%s %s" % (names[i], pct(s[1 + i]).strip()) for i in range(7)))
    print("   its vector == hierarchical predictor %s, == median %s, == left neighbour's result %s" % (pct(s[8]), pct(s[9]), pct(s[10])))
    print("   distinct vectors among the seven: " + ", ".join("%d: %s" % (i + 1, pct(s[11 + i]).strip()) for i in range(7)))
    print("   r4 speculation: left in {zero, global, hier, up, ahead} %s, median in it %s, both %s;  predictor phase ends on up %s, ahead %s, hier %s, any of the five %s" % tuple(pct(s[21 + i]) for i in (7, 8, 0, 1, 2, 3, 4)))
    print("   r4 speculation: whole predictor phase + refinement centre known before the walk: centre == up %s, centre in the five %s" % (pct(s[26]), pct(s[27])))
    print("   first hexagon: a point wins in %s of %d tries;  rescue of a bad block: %s" % ("%.1f %%" % (100.0 * s[18] / max(1, s[19])), int(s[19]), pct(s[20])))
