"""developer tool: is the Degrain cell kernel's FETCH_SIZE excess real over-fetch or an artefact of the x2 counter
correction?  Runs Super + Analyse + Degrain3 of a 4K16 batch on (a) the bench clip (content moves (+3,-1) px per frame: vectors of +-6 /
+-12 / +-18 half-pel units, i.e. compensated rows at 2-byte-odd sample positions half of the time in y-odd cases) and (b) a clip of identical frames
+ noise (all vectors ~0: every compensated row dword-aligned, all four covering blocks read the same lines).  Run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/degrain_traffic.py moving|static [batch]
and compare FETCH_SIZE of degrain_cell_kernel with the algorithmic bytes printed here."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
import mvtools_amd as mv

kind = sys.argv[1] if len(sys.argv) > 1 else "moving"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = bench.CONFIGS["cfg3"]
w, h, bits, tr = cfg[0], cfg[1], cfg[2], cfg[3]
dev = torch.device("cuda", 0)
n = B + 2 * tr
src = bench.synth_clip_device(torch, w, h, bits, n, 7, dev)
if kind == "static":  # every frame = frame 0 + its own +-2 LSB (8-bit scale) noise
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    base = [p.view(torch.int16).to(torch.int32) & 0xffff for p in src[0]]
    for f in range(1, n):
        for p in range(3):
            v = (base[p] + torch.randint(-512, 513, base[p].shape, generator=g, device=dev, dtype=torch.int32)).clamp(0, 65535)
            src[f][p].view(torch.int16)[...] = torch.where(v > 32767, v - 65536, v).to(torch.int16)
p = bench.Pipeline(mv, torch, cfg, B, dev, 0, src=src)
p.step()
torch.cuda.synchronize()
luma = w * h * 2
print("%s clip, batch %d: algorithmic bytes per launch  luma kernel %.2f GB (src + 6 refs + out), chroma kernel %.2f GB" % (
    kind, B, B * luma * 8 / 1e9, B * luma / 2 * 8 / 1e9))
