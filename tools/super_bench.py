#!/usr/bin/env python3
"""mv.Super alone on resident frames: milliseconds per frame batch and HBM bytes/s by variant (sharp, shadow planes, rows kernels).
    python tools/super_bench.py [width height bits frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vapoursynth-mvtools_amd"))
import torch  # noqa: E402
import mvtools_amd as mv  # noqa: E402

w, h, bits, n = (int(a) for a in (sys.argv[1:5] + ["3840", "2160", "16", "96"][len(sys.argv) - 1:]))
bps = 1 if bits <= 8 else 2
pitch = [(w * bps + 255) // 256 * 256, (w // 2 * bps + 255) // 256 * 256, (w // 2 * bps + 255) // 256 * 256]
frames = mv.arena_frames(n, [(h, pitch[0]), (h // 2, pitch[1]), (h // 2, pitch[2])], zero=False)
for f in frames:
    for p in f:
        p.random_(0, 256)
src_bytes = (w * h + 2 * (w // 2) * (h // 2)) * bps


def run(label, rows_off=0, **kw):
    mv.debug_option("super_rows_off", rows_off)
    sup = mv.Super(w, h, bits, **kw)
    out = sup.alloc(n)
    sup.build(frames, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sup.build(frames, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    i = sup.info
    super_bytes = sum(i.plane_width[p] * bps * i.plane_height[p] for p in range(sup.nplanes))
    total = src_bytes + super_bytes * (2 if sup.shadow else 1)
    print("%-34s %7.2f ms / %d frames   %6.2f TB/s (source read once + super%s written once)" % (label, dt * 1e3, n, total * n / dt / 1e12, " + shadow planes" if sup.shadow else ""))
    del out


run("sharp 2, shadows, rows kernels")
run("sharp 2, shadows, tile kernels", rows_off=1)
run("sharp 2, no shadows, rows", shadow=False)
run("sharp 2, no shadows, tile", rows_off=1, shadow=False)
run("sharp 0, shadows, rows", sharp=0)
run("sharp 0, no shadows, rows", sharp=0, shadow=False)
run("sharp 1, shadows, rows", sharp=1)
run("pel 1, shadows", pel=1)
mv.debug_option("super_rows_off", 0)
