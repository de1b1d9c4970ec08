#!/usr/bin/env python3
"""bench.py -- end-to-end mv.Super + mv.Analyse(x2*tr) + mv.DegrainN throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|cfg1|cfg5] [--batch B] [--no-cpu]

A "step" is one pass of the hot path over one batch of B synthetic frames already resident in HBM:
Super of the B+2*tr frames (batch + temporal halo), 2*tr*B motion searches (one chain per frame and direction, all in
one launch per vector clip), DegrainN of the B frames.  value = frames/s over all ranks (frames are independent, so
N GPUs = N disjoint frame ranges, no data-path collective: weak scaling).  Prints ONE JSON line on rank 0.

--gpus N: one process per GPU.  Started by a launcher (WORLD_SIZE set, e.g. python -m torch.distributed.run) the process
is one of N ranks and N must equal WORLD_SIZE; started directly with N > 1 it launches the N ranks itself through
torch.distributed.run on 127.0.0.1 and fails loudly when the node has fewer than N GPUs.  The ranks' frame ranges come
from mvtools_amd.shard.RankPlan (the clip is N*B output frames plus a tr-frame lead-in / lead-out; rank r owns a
contiguous range and holds its tr-frame halo, which it runs Super on itself).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CONFIGS = {
    # name: (width, height, bits, radius, analyse kwargs, super kwargs, default batch, BASELINE.json config string)
    "cfg1": (640, 360, 8, 1, dict(blksize=8), dict(pel=1), 1536, "640x360 YUV420P8 Degrain1 blksize=8 pel=1"),
    # (default batches: the number of chains = 2 * radius * batch decides how many chains share a SIMD -- cfg3: 2046 chains = two per
    # SIMD in one round (the speculative kernel's 256-register build: r4), cfg2: 4096 = four per SIMD, cfg5: 2016 = two per SIMD, all its
    # 1 GB super frames leave room for)
    "cfg2": (1920, 1080, 8, 1, dict(blksize=8, overlap=4, search=4), dict(pel=2), 2048, "1080p YUV420P8 Degrain1 blksize=8 overlap=4 pel=2 search=4"),
    "cfg3": (3840, 2160, 16, 3, dict(blksize=16, overlap=8), dict(pel=2), 341, "4K YUV420P16 Degrain3 blksize=16 overlap=8 pel=2"),
    "cfg5": (7680, 4320, 16, 6, dict(blksize=32, overlap=16), dict(pel=2), 168, "8K YUV420P16 Degrain6 blksize=32 overlap=16 pel=2"),
    # (not a BASELINE configuration: the common HD setting of the reference's users -- 8-bit, 16x16 blocks overlapping by half; r5, `other_configs`)
    "hd16": (1920, 1080, 8, 1, dict(blksize=16, overlap=8), dict(pel=2), 2048, "1080p YUV420P8 Degrain1 blksize=16 overlap=8 pel=2"),
    "hd16l": (1920, 1080, 8, 1, dict(blksize=16, overlap=8, chroma=0), dict(pel=2), 2048, "1080p YUV420P8 Degrain1 blksize=16 overlap=8 pel=2, luma-only search (chroma=0)"),
    "hd16s": (1920, 1080, 8, 1, dict(blksize=16), dict(pel=2), 2048, "1080p YUV420P8 Degrain1 blksize=16 overlap=0 pel=2"),  # (blocks side by side: the reference's default overlap)
    # BASELINE config 4: frame-rate conversion instead of denoising (radius field = 0 selects PipelineFPS)
    "cfg4": (1920, 1080, 8, 0, dict(blksize=8), dict(pel=2), 2047, "1080p YUV420P8 Compensate + BlockFPS 24->60 blksize=8 pel=2"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0  # the same guide's measured copy ceiling


def synth_clip_device(torch, width, height, bits, nframes, seed, device, first_frame=0, total_frames=None):
    """Textured 4:2:0 clip generated on the device: band-limited texture + 8x8 checker translating (+3,-1) px/frame,
    a rectangle of different texture moving (-2,+2), +-2 LSB (8-bit scale) noise.  SURVEY.md 8(d).
    Frame f of the result is frame first_frame + f of ONE clip of total_frames frames: texture, rectangle and noise are functions of the
    GLOBAL frame index (and the seed) only, so every rank of a sharded run generates identical bytes for the frames it shares with its
    neighbours (its tr-frame halo) -- r5; checked across ranks by main()'s shard_check."""
    total = nframes if total_frames is None else total_frames
    scale = 1 << (bits - 8)
    pm = (1 << bits) - 1
    margin = 32 + 4 * total

    def tex(p, y0, x0, h, w, s):
        """plane p's texture at canvas rows y0, y0 + s, ... / columns x0, x0 + s, ... (an analytic function of the canvas position)"""
        yy = torch.arange(y0, y0 + h * s, s, device=device, dtype=torch.float32)[:, None]
        xx = torch.arange(x0, x0 + w * s, s, device=device, dtype=torch.float32)[None, :]
        if p == 0:
            return (40 * torch.sin(xx * 0.21 + yy * 0.07) + 30 * torch.sin(xx * 0.05 - yy * 0.13) + 20 * torch.sin(xx * 0.33 + 1.3) * torch.cos(yy * 0.27)
                    + 25 * (((xx.long() // 8) + (yy.long() // 8)) & 1).float() + 120)
        return 20 * torch.sin(xx * 0.11 + yy * 0.05 + (0.3, 1.7)[p - 1]) + 128
    import mvtools_amd as mv
    shapes = []
    for p in range(3):
        s_ = 2 if p else 1
        rowbytes = (width // s_) * (2 if bits > 8 else 1)
        shapes.append((height // s_, (rowbytes + 255) // 256 * 256))
    arena = mv.arena_frames(nframes, shapes, device)  # one allocation for the whole clip (see Super.alloc)
    g = torch.Generator(device=device)
    frames = []
    for fl in range(nframes):
        f = first_frame + fl
        g.manual_seed(seed * 1000003 + f)               # the frame's noise depends on its global index only
        ox, oy = margin + 3 * f, margin - 1 * f
        planes = []
        for p in range(3):
            s = 2 if p else 1
            h, w = height // s, width // s
            img = tex(p, oy, ox, h, w, s)
            per = max(8, min(height * 5 // 24, width // 4) - 8)   # the rectangle bounces so that long clips keep it inside
            fb = f % (2 * per)
            fb = fb if fb < per else 2 * per - fb
            rx, ry = (width // 2 - 2 * fb) // s, (height // 3 + 2 * fb) // s
            rw, rh = (width // 4) // s, (height // 4) // s
            img[ry:ry + rh, rx:rx + rw] = tex(p, 8, 8, rh, rw, s) * 0.8 + (35 if p == 0 else 10)
            img = img + torch.randint(-2, 3, img.shape, generator=g, device=device).float()
            v = torch.clamp(torch.round(img * scale), 0, pm).to(torch.int32)
            rowbytes = w * (2 if bits > 8 else 1)
            pitch = (rowbytes + 255) // 256 * 256
            t = arena[fl][p]
            assert t.shape == (h, pitch)
            if bits > 8:
                t[:, :rowbytes] = torch.stack([(v & 0xFF), (v >> 8)], dim=-1).to(torch.uint8).reshape(h, rowbytes)
            else:
                t[:, :rowbytes] = v.to(torch.uint8)
            planes.append(t)
        frames.append(planes)
    return frames


def _order_behind_caller(torch, stream, device):
    """A pipeline computes on its own HIP stream, which torch creates non-blocking: NOT ordered behind the stream the caller
    allocated, zero-filled and generated the clip on.  Everything the caller has enqueued so far (clip generation, the zero fill of the
    super / blob arenas, a new `src`) must be ahead of the pipeline's kernels: one event wait, no host synchronisation."""
    if os.environ.get("MVX_BENCH_NO_STREAM_ORDER") == "1":  # evidence only (tools/stress_sharding.py): round 4's behaviour, which raced
        return
    stream.wait_stream(torch.cuda.current_stream(device))


class Pipeline:
    """Super -> Analyse x 2tr -> DegrainN over a resident batch, on the pipeline's own HIP stream (ordered behind the caller's
    stream at construction and at every step; the caller synchronises -- or waits on `stream` -- before it reads the results)."""

    def __init__(self, mv, torch, cfg, batch, device, seed, src=None, plan=None):
        (self.w, self.h, self.bits, self.tr, akw, skw, _, self.label) = cfg
        self.mv, self.torch, self.B, self.device = mv, torch, batch, device
        tr = self.tr
        from mvtools_amd import shard
        # the rank's share of the clip: which frames it outputs, which it holds (share + tr halo), and its job tables
        self.plan = plan if plan is not None else shard.RankPlan(batch + 2 * tr, 0, 1, tr, first_out=tr, last_out=batch + tr)
        assert len(self.plan.outputs()) == batch
        self.n = self.plan.held[1] - self.plan.held[0]
        # `src` lets a second pipeline slot (own filter handles, own super / vector / output buffers, own stream) share
        # the read-only input clip
        self.src = src if src is not None else synth_clip_device(torch, self.w, self.h, self.bits, self.n, seed, device,
                                                                 first_frame=self.plan.held[0], total_frames=self.plan.num_frames)
        self.stream = torch.cuda.Stream(device=device)
        self.sup = mv.Super(self.w, self.h, self.bits, **skw)
        self.supers = self.sup.alloc(self.n, device=device)
        self.an = {}
        for d in range(1, tr + 1):
            for isb in (1, 0):
                self.an[(d, isb)] = mv.Analyse(self.sup, isb=isb, delta=d, **akw)
        a0 = self.an[(1, 1)]
        self.blobs = {k: a.alloc_blobs(batch, device=device) for k, a in self.an.items()}
        self.dg = mv.Degrain(tr, self.sup, a0.ad, [p.stride(0) for p in self.src[0]])
        self.out = mv.arena_frames(batch, [tuple(p.shape) for p in self.src[0]], device, zero=False)
        self.ev = []  # (start, end) events around the search launches
        # several batches in flight (--slots): every slot has its own stream.  A search launch of 2046 chains fills the GPU's wave slots (two waves per SIMD at 256
        # registers), so the next slot's launch is dispatched chain by chain as the current one's chains finish -- its tail is filled --, while Super of the
        # next batch and Degrain of the previous one run under the running search
        _order_behind_caller(torch, self.stream, device)

    def step(self, time_search=False, src=None):
        _order_behind_caller(self.torch, self.stream, self.device)
        with self.torch.cuda.stream(self.stream):
            self._step(time_search, self.src if src is None else src)

    def _step(self, time_search, src):
        torch, tr, B = self.torch, self.tr, self.B
        self.sup.build(src, out=self.supers)
        jobs, blobs = self._search_jobs()
        if time_search:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.an[(1, 1)].run(jobs, blobs=blobs)
        if time_search:
            e1.record()
            self.ev.append((e0, e1))
        djobs = []
        for n, refs, i in self.plan.degrains():
            djobs.append((src[n], [self.supers[r] if r is not None else None for r in refs], [self.blobs[key][i] for key in self.plan.clips]))
        self.dg.run(djobs, out=self.out)

    def _search_jobs(self):
        """all 2*tr vector clips share one parameter block (delta / isb only pick the reference frame), so every chain of the step goes into ONE
        launch: 2*tr*B chains resident at once.  -> (jobs, blobs) of that launch"""
        jobs, blobs = [], []
        for key, pairs in self.plan.searches().items():
            jobs += [(self.supers[n], self.supers[nref] if nref is not None else None) for n, nref in pairs]
            blobs += self.blobs[key]
        return jobs, blobs

    def search_alone(self):
        """ONE search launch of this slot with nothing else on the GPU, by HIP events (after the timed region: with several batches in flight the launches of
        the timed steps run with the neighbours' Super / Degrain kernels beside them and take longer): milliseconds"""
        torch = self.torch
        torch.cuda.synchronize()
        with torch.cuda.stream(self.stream):
            jobs, blobs = self._search_jobs()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.an[(1, 1)].run(jobs, blobs=blobs)
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    def level_grids(self):
        """[(nBlkX, nBlkY)] of every level of the search, finest first (GroupOfPlanes.c:25-56)"""
        ad = self.an[(1, 1)].ad
        wb = (ad.nBlkSizeX - ad.nOverlapX) * ad.nBlkX + ad.nOverlapX
        hb = (ad.nBlkSizeY - ad.nOverlapY) * ad.nBlkY + ad.nOverlapY
        return [(((wb >> i) - ad.nOverlapX) // (ad.nBlkSizeX - ad.nOverlapX), ((hb >> i) - ad.nOverlapY) // (ad.nBlkSizeY - ad.nOverlapY)) for i in range(ad.nLvCount)]

    def algorithmic_bytes_per_chain(self):
        """one chain reads the current frame's pyramid (sub-pel plane 0 of every level), the whole
        reference super frame once, and writes the vector blob."""
        i = self.sup.info
        bps = self.sup.bps
        import mvtools_amd as mv
        full, pyr0 = 0, 0
        import ctypes as C
        L = mv.lib()
        for p in range(i.num_planes):
            xr, yr = (i.xRatioUV, i.yRatioUV) if p else (1, 1)
            for lv in range(i.levels):
                # level dims (MVFrame.cpp:1209-1226)
                wl, hl = i.width, i.height
                for _ in range(lv):
                    wl = ((wl // i.xRatioUV + 1) // 2) * i.xRatioUV if i.hpad >= i.xRatioUV else ((wl // i.xRatioUV) // 2) * i.xRatioUV
                    hl = ((hl // i.yRatioUV + 1) // 2) * i.yRatioUV if i.vpad >= i.yRatioUV else ((hl // i.yRatioUV) // 2) * i.yRatioUV
                pw, ph = wl // xr + 2 * (i.hpad // xr), hl // yr + 2 * (i.vpad // yr)
                npl = i.pel * i.pel if lv == 0 else 1
                full += pw * ph * bps * npl
                pyr0 += pw * ph * bps
        return pyr0 + full + self.an[(1, 1)].blob_size, full


class PipelineFPS:
    """BASELINE config 4: Super -> Analyse (backward + forward, delta 1) -> Compensate (every frame towards its successor) +
    BlockFPS 24 -> 60 over a resident batch of B + 1 input frames; a step produces B compensated frames and (B + 1) * 60 / 24
    interpolated ones.  The unit of `value` is interpolated output frames per second."""

    def __init__(self, mv, torch, cfg, batch, device, seed, src=None, plan=None):
        (self.w, self.h, self.bits, _, akw, skw, _, self.label) = cfg
        self.mv, self.torch, self.B, self.device = mv, torch, batch, device
        self.tr = 1
        self.n = batch + 1
        self.plan = plan
        self.src = src if src is not None else synth_clip_device(torch, self.w, self.h, self.bits, self.n, seed, device,
                                                                 first_frame=plan.held[0] if plan is not None else 0,
                                                                 total_frames=plan.num_frames if plan is not None else None)
        self.stream = torch.cuda.Stream(device=device)
        self.sup = mv.Super(self.w, self.h, self.bits, **skw)
        self.supers = self.sup.alloc(self.n, device=device)
        self.an = {(1, 1): mv.Analyse(self.sup, num_frames=self.n, isb=1, **akw), (1, 0): mv.Analyse(self.sup, num_frames=self.n, isb=0, **akw)}
        self.blobs = {k: a.alloc_blobs(self.n, device=device) for k, a in self.an.items()}
        pitch = [p.stride(0) for p in self.src[0]]
        self.comp = mv.Compensate(self.sup, self.an[(1, 1)].ad, dst_pitch=pitch)
        self.fps = mv.BlockFPS(self.sup, self.an[(1, 1)].ad, self.an[(1, 0)].ad, self.n, pitch, 24, 1, num=60, den=1)
        self.nout = self.fps.num_frames
        shapes = [tuple(p.shape) for p in self.src[0]]
        self.comp_out = mv.arena_frames(batch, shapes, device, zero=False)
        self.fps_out = mv.arena_frames(self.nout, shapes, device, zero=False)
        self.ev = []
        self.frames_per_step = self.nout
        _order_behind_caller(torch, self.stream, device)

    def step(self, time_search=False):
        _order_behind_caller(self.torch, self.stream, self.device)
        with self.torch.cuda.stream(self.stream):
            torch, n = self.torch, self.n
            self.sup.build(self.src, out=self.supers)
            jobs = [(self.supers[i], self.supers[i + 1] if i + 1 < n else None) for i in range(n)] + [(self.supers[i], self.supers[i - 1] if i >= 1 else None) for i in range(n)]
            blobs = self.blobs[(1, 1)] + self.blobs[(1, 0)]
            if time_search:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.an[(1, 1)].run(jobs, blobs=blobs)  # both vector clips share one parameter block: one launch, 2 * (B + 1) chains
            if time_search:
                e1.record()
                self.ev.append((e0, e1))
            self.comp.run([(self.supers[i], self.supers[i + 1], self.blobs[(1, 1)][i]) for i in range(self.B)], out=self.comp_out)
            self.fps.run(list(range(self.nout)), self.src, self.supers, self.blobs[(1, 1)], self.blobs[(1, 0)], out=self.fps_out)

    algorithmic_bytes_per_chain = Pipeline.algorithmic_bytes_per_chain
    level_grids = Pipeline.level_grids


def _frame_to_numpy(mv, frame, w, h, bits):
    import numpy as np
    dt = np.uint16 if bits > 8 else np.uint8
    return [mv.plane_to_numpy(frame[p], w >> (1 if p else 0), dt) for p in range(3)]


def shard_payload(mv, torch, cfg, pipe, plan, rank, threads, oracle=True):
    """One rank's contribution to shard_check: checksums of the source frames at both ends of the range it holds -- the frames it shares with its
    neighbours -- and, on ranks 0 and 1, the comparison of ONE output frame at the shard boundary with the oracle (rank 0: its last output frame,
    rank 1: its first; both need the halo)."""
    res = {"rank": rank, "held": list(plan.held), "out": list(plan.out), "sums": {}, "parity": None}
    try:
        tr = max(plan.tr, 1)
        lo, hi = plan.held
        for n in sorted(set(range(lo, min(hi, lo + 2 * tr))) | set(range(max(lo, hi - 2 * tr), hi))):
            acc = []
            for t in pipe.src[n - lo]:
                x = t.reshape(-1).to(torch.int64)
                acc += [int(x.sum().item()), int((x * (torch.arange(x.numel(), device=x.device, dtype=torch.int64) % 65521 + 1)).sum().item())]
            res["sums"][n] = acc
        if oracle and rank in (0, 1) and len(plan.outputs()) > 0:
            _, par = oracle_leg(mv, torch, cfg, pipe, threads, 1, i0=(pipe.B - 1 if rank == 0 else 0))
            res["parity"] = {"rank": rank, "global_frame": plan.out[1] - 1 if rank == 0 else plan.out[0], "identical": par["identical"], "mismatches": par.get("mismatches")}
    except Exception as e:  # (a failed side check must neither cost the headline line nor leave the other ranks waiting in the gather)
        res["error"] = "%s: %s" % (type(e).__name__, e)
    return res


def shard_verdict(gathered):
    """rank 0: every global frame index that two ranks hold must carry identical bytes on both; the output ranges must partition the job"""
    seen, compared, bad = {}, 0, []
    for g in gathered:
        for n, v in g["sums"].items():
            if n in seen:
                compared += 1
                if seen[n][1] != v:
                    bad.append("frame %d: ranks %d and %d hold different bytes" % (n, seen[n][0], g["rank"]))
            else:
                seen[n] = (g["rank"], v)
    covered = sorted(n for g in gathered for n in range(g["out"][0], g["out"][1]))
    res = {"shared_source_frames_compared": compared, "shared_source_frames_identical": not bad, "mismatches": bad[:8],
           "output_ranges_partition_the_job": bool(covered) and covered == list(range(covered[0], covered[0] + len(covered))),
           "boundary_frames_vs_oracle": [g["parity"] for g in gathered if g.get("parity") is not None]}
    errs = [g["error"] for g in gathered if g.get("error")]
    if errs:
        res["errors"] = errs[:4]
    return res


def shard_check(mv, torch, dist, cfg, pipe, plan, rank, world, threads):
    """--gpus N > 1, outside the timed region, control plane only (all_gather_object; no data-path collective).  Returns the dict rank 0 puts
    into the JSON line (None elsewhere)."""
    gathered = [None] * world
    dist.all_gather_object(gathered, shard_payload(mv, torch, cfg, pipe, plan, rank, threads))
    return shard_verdict(gathered) if rank == 0 else None


def oracle_leg(mv, torch, cfg, pipe, threads, F, i0=None):
    """The CPU oracle (scalar C restatement of the reference, kind 'port') on F output frames OF THE TIMED STEP: the clip
    frames they need are downloaded from the device clip, the oracle runs Super / Analyse x 2tr / DegrainN on them with
    `threads` worker threads each owning whole frames (VapourSynth fmParallel style; the wall time of this part is the CPU
    baseline), and every vector blob and every output plane of those frames is compared byte for byte with what the GPU
    left in HBM after the last timed step.  Returns (seconds, parity dict)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import mvoracle as mo
    (w, h, bits, tr, akw, skw, _, label) = cfg
    B = pipe.B
    F = max(1, min(F, B))
    if i0 is None:
        i0 = max(0, B // 2 - F // 2)                   # F consecutive output frames from the middle of the batch (i0 given: from there)
    dgs = pipe.plan.degrains()[i0:i0 + F]              # (local frame, local refs per vector clip, blob index)
    need = sorted({n for n, refs, _ in dgs} | {r for _, refs, _ in dgs for r in refs if r is not None})
    torch.cuda.synchronize()
    host = {n: _frame_to_numpy(mv, pipe.src[n], w, h, bits) for n in need}
    sup = mo.Super(w, h, bits, **skw)
    ans = {(d, isb): mo.Analyse(sup, isb=isb, delta=d, **akw) for d, isb in pipe.plan.clips}
    dg = mo.Degrain(tr, sup, ans[pipe.plan.clips[0]].ad)
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        supers = dict(zip(need, ex.map(lambda n: sup.frame(host[n]), need)))

        def one(job):
            n, refs, _ = job
            blobs = [ans[key].frame(supers[n], supers[r] if r is not None else None) for key, r in zip(pipe.plan.clips, refs)]
            return blobs, dg.frame(host[n], [supers[r] if r is not None else None for r in refs], blobs)
        res = list(ex.map(one, dgs))
    dt = time.time() - t0
    bad = []
    for (n, refs, i), (blobs, out) in zip(dgs, res):
        for key, ob in zip(pipe.plan.clips, blobs):
            if not np.array_equal(pipe.blobs[key][i].cpu().numpy(), ob):
                bad.append("vectors frame %d clip delta=%d isb=%d" % (n, key[0], key[1]))
        for p in range(3):
            if not np.array_equal(mv.plane_to_numpy(pipe.out[i][p], out[p].shape[1], out[p].dtype), out[p]):
                bad.append("Degrain%d output frame %d plane %d" % (tr, n, p))
    parity = {"frames": len(dgs), "vector_blobs": len(dgs) * 2 * tr, "output_planes": 3 * len(dgs), "identical": not bad,
              "against": "oracle/ (CPU restatement) on the same clip frames, downloaded from the device clip of the timed step",
              "first_output_frame_in_batch": i0}
    if bad:
        parity["mismatches"] = bad[:8]
    sample = ("%d output frames of %s taken from the timed step's own clip (%d Super + %d Analyse + %d Degrain%d), %d threads each owning whole frames, "
              "%.1f s wall; scalar C oracle (-O2 -mavx2), not the reference's SIMD build (BASELINE.md 4)") % (
        len(dgs), label, len(need), 2 * tr * len(dgs), len(dgs), tr, threads, dt)
    return {"value": len(dgs) / dt, "unit": "fps", "cores": threads, "kind": "port", "sample": sample}, parity


def oracle_leg_fps(mv, torch, cfg, pipe, threads, F):
    """cfg4 on the host cores: the oracle's Super / Analyse x2 / Compensate / BlockFPS 24 -> 60 on the first F + 1 input frames of
    the timed step's own clip; blobs, compensated and interpolated frames that do not depend on frames beyond the sample are
    compared byte for byte with the GPU's."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import mvoracle as mo
    (w, h, bits, _, akw, skw, _, label) = cfg
    F = max(2, min(F, pipe.B))
    n = F + 1
    torch.cuda.synchronize()
    frames = [_frame_to_numpy(mv, pipe.src[i], w, h, bits) for i in range(n)]
    sup = mo.Super(w, h, bits, **skw)
    abw, afw = mo.Analyse(sup, num_frames=n, isb=1, **akw), mo.Analyse(sup, num_frames=n, isb=0, **akw)
    comp = mo.Compensate(sup, abw.ad)
    fps = mo.BlockFPS(sup, abw.ad, afw.ad, n, 24, 1, num=60, den=1)
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        supers = list(ex.map(sup.frame, frames))
        bbw = list(ex.map(lambda i: abw.frame(supers[i], supers[i + 1] if i + 1 < n else None), range(n)))
        bfw = list(ex.map(lambda i: afw.frame(supers[i], supers[i - 1] if i >= 1 else None), range(n)))
        oc = list(ex.map(lambda i: comp.frame(supers[i], supers[i + 1], bbw[i]), range(F)))
        of = list(ex.map(lambda k: fps.frame(k, frames, supers, bbw, bfw), range(fps.num_frames)))
    dt = time.time() - t0
    bad = []
    for i in range(F):
        if not np.array_equal(pipe.blobs[(1, 1)][i].cpu().numpy(), bbw[i]):
            bad.append("backward vectors frame %d" % i)
        if not np.array_equal(pipe.blobs[(1, 0)][i].cpu().numpy(), bfw[i]):
            bad.append("forward vectors frame %d" % i)
        for p in range(3):
            if not np.array_equal(mv.plane_to_numpy(pipe.comp_out[i][p], oc[i][p].shape[1], oc[i][p].dtype), oc[i][p]):
                bad.append("Compensate frame %d plane %d" % (i, p))
    checked = 0
    for k in range(fps.num_frames):
        nl, nr, _ = fps.map(k)
        if nr > F or nl >= F:  # (the sample clip ends there: its last backward vectors are invalid, the bench clip's are not)
            continue
        checked += 1
        for p in range(3):
            if not np.array_equal(mv.plane_to_numpy(pipe.fps_out[k][p], of[k][p].shape[1], of[k][p].dtype), of[k][p]):
                bad.append("BlockFPS output frame %d plane %d" % (k, p))
    parity = {"frames": checked, "vector_blobs": 2 * F, "output_planes": 3 * (F + checked), "identical": not bad,
              "against": "oracle/ (CPU restatement) on the same clip frames, downloaded from the device clip of the timed step"}
    if bad:
        parity["mismatches"] = bad[:8]
    sample = ("%d input frames of %s taken from the timed step's own clip -> %d interpolated + %d compensated frames, %d threads each owning whole frames, "
              "%.1f s wall; scalar C oracle (-O2 -mavx2), not the reference's SIMD build (BASELINE.md 4)") % (
        n, label, fps.num_frames, F, threads, dt)
    return {"value": fps.num_frames / dt, "unit": "fps", "cores": threads, "kind": "port", "sample": sample}, parity


def search_kernel_name(mv):
    """which kernel the last search launch of this process ran (mvx_debug_last_launch)"""
    import ctypes as C
    info = (C.c_int * 5)()
    mv.lib().mvx_debug_last_launch(info)
    if info[4] == 2:
        return "analyse_spec_kernel, %d chains per SIMD" % info[0]
    if info[4] == 3:
        return "analyse_spec_kernel (team form: %d waves per chain)" % info[1]
    return "analyse_fast_kernel, %d chains per SIMD" % info[0] if info[0] else "analyse_kernel"


def other_configs():
    """The other BASELINE configurations that run on one GPU, each as a short child run of this script (4 timed steps after 2 warm-ups -- one per batch in flight --, its
    own two-frame comparison against the oracle; the parent has released its device memory): {"cfg2": {...}, ...}.  Outside every timed region."""
    import subprocess
    res = {}
    for c in ("cfg2", "cfg4", "cfg5", "hd16"):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", c, "--steps", "4", "--warmup", "2", "--no-cpu", "--no-traffic", "--no-others"]
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=420)
            line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
            r = d["roofline"]
            res[c] = {"workload": d["config"]["workload"], "fps": d["value"], "ms_per_step": d["ms_per_step"], "frames_per_step": d["config"].get("frames_per_step_per_gpu"),
                      "search_kernel": r.get("kernel"), "launch_ms": r.get("avg_launch_ms"), "frac": r.get("frac"),
                      "parity": bool(d.get("parity_check", {}).get("identical")), "parity_frames": d.get("parity_check", {}).get("frames"), "rc": p.returncode}
        except Exception as e:  # (a failed side run must not cost the headline line)
            res[c] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def vs_shell_leg(frames=640, threads=48):
    """The drop-in boundary in the driver's record (r5): the VapourSynth filter shell (libmvtools_vs.so) in the mini host, the cfg3 graph -- mv.Super ->
    mv.Analyse x 6 -> mv.Degrain3, blksize 16, overlap 8 -- over `frames` 4K16 frames with `threads` request threads (r6: one per logical CPU of the host, at most 256, as a VapourSynth core starts its workers -- 256 on the bench box; the shell's admission gate keeps 96 output frames in flight whatever the host asks for; r5 used 32: MVX_VS_BENCH_THREADS) asking for OUTPUT frames in frame
    order, in a child process of its own (this function IS that child: `bench.py --vs-shell-leg`).  The clip is generated on the device and written to a
    raw file the host reads; `fps_all_inclusive` = frames / wall clock from "clip in host memory" to "last output frame delivered": graph construction
    (which starts the first look-ahead windows), every upload / download over PCIe, and the shell's per-frame work included; `fps_steady` = frames / the
    request phase alone.  The result file is compared plane by plane with the same graph through the batched C ABI (bench.Pipeline over the whole
    clip, missing references at the clip ends exactly as the filters see them).  Modes: the default, and MVX_VS_SUPER_LAZY=1 (mv.Super's pixels stay on
    the device; opt-in).  Prints one JSON object."""
    import re
    import subprocess
    import tempfile
    import numpy as np
    import torch
    import mvtools_amd as mv
    from mvtools_amd import shard
    cfg = CONFIGS["cfg3"]
    (w, h, bits, tr, akw, skw, _, label) = cfg
    here = os.path.join(ROOT, "vapoursynth-mvtools_amd")
    host, plugin = os.path.join(here, "mvx_vs_host"), os.path.join(here, "libmvtools_vs.so")
    if not (os.path.exists(host) and os.path.exists(plugin)):
        return {"error": "mvx_vs_host / libmvtools_vs.so not built"}
    device = torch.device("cuda", 0)
    tmp = tempfile.mkdtemp(prefix="mvx_vs_", dir=os.environ.get("TMPDIR", "/tmp"))
    src, outp = os.path.join(tmp, "in.raw"), os.path.join(tmp, "out.raw")
    res = {"workload": label + " through VapourSynthPluginInit2 / getFrame", "frames": frames, "threads": threads}
    try:
        import gc
        import xxhash
        per = w * h + 2 * (w // 2) * (h // 2)
        plane_dims = [(w, h), (w // 2, h // 2), (w // 2, h // 2)]

        def file_digests(path):
            """xxh64 of every plane of every frame of a raw planar 16-bit clip file, read frame by frame"""
            out = []
            with open(path, "rb") as f:
                for n in range(frames):
                    for pw, ph in plane_dims:
                        out.append(xxhash.xxh64(f.read(pw * ph * 2)).intdigest())
            return out
        # 1. the clip (generated on the device, a function of the frame index) -> a raw file the host reads.  The device is EMPTY when the host runs:
        # a process that allocates right after another one freed ~200 GB waits for the driver to scrub it (the first look-ahead window took 1.7 s
        # instead of 0.1 s when the C-ABI reference ran first: profiles/r5_vs_shell_640frames_window_trace_reference_first.txt)
        clip = synth_clip_device(torch, w, h, bits, frames, 1000, device)
        torch.cuda.synchronize()
        t0 = time.time()
        with open(src, "wb") as f:
            for fr in clip:
                for p, t in enumerate(fr):
                    f.write(t[:, :(w >> (1 if p else 0)) * 2].contiguous().cpu().numpy().tobytes())
        res["clip_file_s"] = round(time.time() - t0, 1)
        del clip
        gc.collect()
        torch.cuda.empty_cache()
        # the host processes are measured on a device in its steady state: (a) the first process after boot that maps ~200 GB pays seconds of
        # driver time in its allocations (r4: "the FIRST process on a fresh box measures 134-147 fps"; r5: 7.8 thread-seconds in the windows' arena
        # allocations, against 0.5 in the next process) -- so this process maps and releases that much once; (b) a process that starts right after
        # another one released ~200 GB waits for the driver to scrub it (2.7 s of graph construction instead of 1.1) -- so every host run starts
        # after a pause.  Neither belongs to the plugin; both are outside every timed interval.
        settle = float(os.environ.get("MVX_VS_SETTLE_S", "4"))
        try:
            free_b = torch.cuda.mem_get_info(device)[0]
            warm = torch.empty(int(free_b * 0.85), dtype=torch.uint8, device=device)
            warm[::4096] = 0
            torch.cuda.synchronize()
            del warm
        except Exception:
            pass
        gc.collect()
        torch.cuda.empty_cache()

        def run(extra_env, key):
            if os.path.exists(outp):
                os.remove(outp)  # (the mini host opens its result file "wb" after it loaded the clip: truncating the previous run's 16 GB took ~1 s of "graph construction")
            time.sleep(settle)
            env = dict(os.environ, MVX_HOST_TIMES="1", GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "16"), **extra_env)
            t0 = time.time()
            r = subprocess.run([host, plugin, "run", "degrain3", src, str(w), str(h), str(bits), str(frames), outp, "a.blksize=16", "a.overlap=8",
                                "x.threads=%d" % threads, "x.order=frame"] + (["x.cache=%s" % os.environ["MVX_VS_BENCH_CACHE"]] if os.environ.get("MVX_VS_BENCH_CACHE") else []),
                               capture_output=True, text=True, env=env, timeout=600)
            d = {"process_wall_s": round(time.time() - t0, 2), "rc": r.returncode}
            if os.environ.get("MVX_VS_KEEP_STDERR"):  # developer: the shell's statistics / window trace (MVX_VS_STATS=1, MVX_VS_TRACE=1) of this run
                open(os.path.join(os.environ["MVX_VS_KEEP_STDERR"], "vs_shell_stderr_%s.txt" % ("lazy" if extra_env else "default")), "w").write(r.stderr)
            if r.returncode != 0 or "DONE" not in r.stdout:
                d["error"] = (r.stderr or r.stdout)[-300:]
                return d, None
            g = lambda pat: float(re.search(pat, r.stderr).group(1))
            loaded, built, req = g(r"clip loaded at ([0-9.]+) s"), g(r"graph built at ([0-9.]+) s"), g(r"output clip \(frame order\) ([0-9.]+) s")
            d.update({"mode": key, "graph_construction_s": round(built - loaded, 2), "request_phase_s": req,
                      "fps_all_inclusive": frames / (built - loaded + req), "fps_steady": frames / req})
            return d, (file_digests(outp) if os.path.getsize(outp) == per * 2 * frames else None)
        if os.environ.get("MVX_VS_LAZY_FIRST"):  # developer: which of the two runs comes first
            lazy, dig_lazy = run({"MVX_VS_SUPER_LAZY": "1"}, "MVX_VS_SUPER_LAZY=1 (mv.Super's pixels stay on the device; opt-in)")
            first, dig_default = run({}, "default (mv.Super delivers its frames to the host)")
        else:
            first, dig_default = run({}, "default (mv.Super delivers its frames to the host)")
            lazy, dig_lazy = run({"MVX_VS_SUPER_LAZY": "1"}, "MVX_VS_SUPER_LAZY=1 (mv.Super's pixels stay on the device; opt-in)")
        # 2. the reference result: the same graph through the batched C ABI (bench.Pipeline over the whole clip: references beyond its ends are missing, MVAnalyse.c:120-129)
        plan = shard.RankPlan(frames, 0, 1, tr)
        pipe = Pipeline(mv, torch, cfg, frames, device, seed=1000, plan=plan)
        pipe.step()
        torch.cuda.synchronize()
        want = []
        for fr in pipe.out:
            for p, t in enumerate(fr):
                want.append(xxhash.xxh64(t[:, :(w >> (1 if p else 0)) * 2].contiguous().cpu().numpy().tobytes()).intdigest())
        for d, dig in ((first, dig_default), (lazy, dig_lazy)):
            if "error" not in d:
                d["identical_to_c_abi"] = dig == want
                if dig is None or dig != want:
                    d["planes_that_differ"] = -1 if dig is None else sum(1 for a, b in zip(dig, want) if a != b)
        res["compared_by"] = "xxh64 of every plane of every frame (%d digests per run)" % len(want)
        res.update(first)
        res["lazy_super"] = lazy
    except Exception as e:  # (a failed side run must not cost the headline line)
        res["error"] = "%s: %s" % (type(e).__name__, e)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def vs_shell():
    """child run of vs_shell_leg (the parent has released its device memory); {"error": ...} on any failure"""
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--vs-shell-leg"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
        return json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def measure_traffic(args, B):
    """HBM bytes of ONE launch of the search kernel, measured now: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace
    only, one counter per pass) over `bench.py --steps 1 --warmup 0` of the same configuration, corrected as MI355X_MICROARCH.md
    prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-byte request of 16-byte-per-lane reads: bytes = 2 * FETCH_SIZE KB + WRITE_SIZE
    KB).  The caller must have released its device memory.  Returns (bytes or None, how it was obtained)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "not measured: rocprofv3 not found"
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mvx_pmc_", dir="/tmp")
        # (counters for the search kernels only: with every torch kernel of the clip generator instrumented too, rocprofv3 7.2 crashed -- SIGSEGV inside a torch
        # multiply -- in two of two r5 runs; the bytes this function reports are the search launch's anyway)
        cmd = [exe, "--pmc", c, "--kernel-trace", "--kernel-include-regex", "analyse_", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--no-cpu", "--no-parity", "--no-traffic", "--steps", "1", "--warmup", "0", "--config", args.config, "--batch", str(B), "--slots", str(args.slots)]
        try:
            pr = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
            if pr.returncode != 0:
                return None, "not measured: rocprofv3 --pmc %s pass exited with %d: %s" % (c, pr.returncode, pr.stderr.decode(errors="replace")[-400:].replace("\n", " | "))
            tot, n = 0.0, 0
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] == c and "analyse" in r["Kernel_Name"] and "divide" not in r["Kernel_Name"]:
                            tot += float(r["Counter_Value"])
                            n += 1
            if n == 0:
                return None, "not measured: no %s rows for the search kernel" % c
            vals[c] = tot / n
        except Exception as e:  # (a profiler failure must not cost the bench line)
            return None, "not measured: rocprofv3 --pmc %s pass failed (%s)" % (c, type(e).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, in this run; 2 * FETCH_SIZE KB + WRITE_SIZE KB (gfx950 correction)"


def ingest_run(torch, pipe, steps, warmup):
    """Secondary, ingest-inclusive measurement (--ingest; DegrainN configurations): every step first receives its source frames from
    PINNED host memory and its output frames go back to pinned host memory, on a copy stream, overlapped with the compute of the
    neighbouring steps (the source clip is double-buffered on the device; the output download of step i runs under Super + search of
    step i + 1 and is waited for before step i + 1's Degrain writes).  Returns seconds per step."""
    dev_a = pipe.src[0][0]._base                              # the clip's arena (mv.arena_frames: one allocation)
    dev_b = torch.empty_like(dev_a)
    off = [[p.storage_offset() for p in f] for f in pipe.src]
    src_b = [[dev_b[o:o + p.numel()].view(p.shape) for p, o in zip(f, of)] for f, of in zip(pipe.src, off)]
    bufs, srcs = [dev_a, dev_b], [pipe.src, src_b]
    out_dev = pipe.out[0][0]._base
    host_src = torch.empty(dev_a.numel(), dtype=torch.uint8).pin_memory()
    host_src.copy_(dev_a)
    host_out = torch.empty(out_dev.numel(), dtype=torch.uint8).pin_memory()
    cs = torch.cuda.Stream(device=dev_a.device)
    torch.cuda.synchronize()
    up_ev, done_ev, dl_ev = {}, None, None

    def upload(i):
        with torch.cuda.stream(cs):
            bufs[i % 2].copy_(host_src, non_blocking=True)
            up_ev[i] = cs.record_event()
    t0 = None
    upload(0)
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if i + 1 < warmup + steps:
            if done_ev is not None:
                cs.wait_event(done_ev)                        # (the buffer being refilled was read by step i - 1)
            upload(i + 1)
        pipe.stream.wait_event(up_ev[i])
        if dl_ev is not None:
            pipe.stream.wait_event(dl_ev)                     # (conservative: the whole step waits for the previous download, not only its Degrain)
        pipe.step(src=srcs[i % 2])
        done_ev = pipe.stream.record_event()
        with torch.cuda.stream(cs):
            cs.wait_event(done_ev)
            host_out.copy_(out_dev, non_blocking=True)
            dl_ev = cs.record_event()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, dev_a.numel(), out_dev.numel()


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) through torch.distributed.run, the
    same way the driver does, and hand back its exit status.  Fails loudly when the node has fewer than N GPUs."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py: --gpus %d requested but this node exposes %d GPU(s); refusing to run fewer ranks than asked for\n" % (n, have))
        return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def world_from_env(gpus, env=None):
    """(rank, local_rank, world) of this process; the launcher's WORLD_SIZE must agree with --gpus"""
    env = os.environ if env is None else env
    world = int(env.get("WORLD_SIZE", "1"))
    if world != gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world))
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--slots", type=int, default=0, help="batches in flight: slot i owns its buffers and HIP stream, so the Super kernels of the next batch and the "
                    "Degrain kernels of the previous one run under the (latency-bound) search of the current one, and the next search launch fills the wave slots "
                    "the current one's finishing chains free.  Default: 2 (cfg3 +11 %%: profiles/r5_batches_in_flight_unchained.txt, 2 x 107 GB of the 288 GB; "
                    "cfg2 / cfg4 / hd16 +3 %%: their chains finish together, there is little tail to fill), 1 for cfg5 (no room).  Stream priorities change nothing "
                    "(profiles/r5_batches_in_flight_other_configs.txt)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline (a two-frame parity check of the timed step still runs)")
    ap.add_argument("--no-parity", action="store_true", help="with --no-cpu: skip the oracle comparison of the timed step too")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure the search launch's HBM traffic")
    ap.add_argument("--no-others", action="store_true", help="default cfg3 run: do not add the short runs of cfg2 / cfg4 / cfg5 (`other_configs` of the JSON line)")
    ap.add_argument("--no-vs", action="store_true", help="default cfg3 run: do not add the run of the graph through the VapourSynth filter shell (`vs_shell` of the JSON line)")
    ap.add_argument("--vs-shell-leg", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--vs-frames", type=int, default=640)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--ingest", action="store_true", help="also measure the step with its source frames arriving from / its output frames leaving to pinned host memory "
                    "(secondary metric `ingest_inclusive` of the JSON line; `value` stays the resident-input number)")
    args = ap.parse_args()

    if args.vs_shell_leg:
        print(json.dumps(vs_shell_leg(args.vs_frames, int(os.environ.get("MVX_VS_BENCH_THREADS", str(max(8, min(os.cpu_count() or 48, 256))))))))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank, local_rank, world = world_from_env(args.gpus)

    import torch
    import mvtools_amd as mv
    from mvtools_amd import shard
    mv.lib()
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (local rank %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    cfg = CONFIGS[args.config]
    if args.slots <= 0:
        args.slots = 1 if args.config == "cfg5" else 2
    B = args.batch or cfg[6]
    tr = cfg[3]
    fpsconv = tr == 0  # cfg4: Compensate + BlockFPS instead of DegrainN
    PipeT = PipelineFPS if fpsconv else Pipeline
    # the job: world*B output frames of one clip with a tr-frame lead-in / lead-out; this rank's contiguous share + halo
    plan = shard.RankPlan(world * B + 2 * max(tr, 1), rank, world, max(tr, 1), first_out=max(tr, 1), last_out=world * B + max(tr, 1))
    # ONE clip for the whole job: every rank generates the frames it holds (its share + the tr-frame halo) from the same seed; the content of a
    # frame is a function of its global index, so neighbouring ranks hold identical copies of the frames they share (shard_check below)
    pipe = PipeT(mv, torch, cfg, B, device, seed=1000, plan=plan)
    pipes = [pipe] + [PipeT(mv, torch, cfg, B, device, seed=1000, src=pipe.src, plan=plan) for _ in range(max(1, args.slots) - 1)]
    units = pipe.frames_per_step if fpsconv else B  # frames a step delivers
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        pipes[i % len(pipes)].step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):  # exactly K steps = K batches, round-robin over the slots
        pipes[i % len(pipes)].step(time_search=True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    rc = 0
    shard = None
    if dist is not None and not fpsconv and not args.no_parity:  # every rank takes part (control-plane gather, after the timed region)
        try:
            shard = shard_check(mv, torch, dist, cfg, pipes[(args.steps - 1) % len(pipes)] if args.steps else pipe, plan, rank, world, max(1, min((os.cpu_count() or 8) // world, 16)))
        except Exception as e:
            shard = {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None
    if rank == 0:
        search_ms = [a.elapsed_time(b) for pp in pipes for a, b in pp.ev]
        avg_launch_ms = sum(search_ms) / len(search_ms)
        # with several batches in flight consecutive search launches OVERLAP (the next one fills the slots the current one frees): their event-to-event durations add
        # up to more than the time the GPU spent searching.  The kernel's time per launch is then the length of the UNION of the launches' intervals / launches
        evs = [e for pp in pipes for e in pp.ev]
        iv = sorted((evs[0][0].elapsed_time(a), evs[0][0].elapsed_time(b)) for a, b in evs)
        busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
        for s_, e_ in iv[1:]:
            if s_ > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += cur_e - cur_s
        busy_launch_ms = busy / len(evs)
        overlapped = len(pipes) > 1 and busy_launch_ms < 0.98 * avg_launch_ms
        event_launch_ms = avg_launch_ms
        if overlapped:
            avg_launch_ms = busy_launch_ms
        bytes_chain, full = pipe.algorithmic_bytes_per_chain()
        chains = 2 * (B + 1) if fpsconv else 2 * cfg[3] * B
        achieved = bytes_chain * chains / (avg_launch_ms * 1e-3) / 1e9
        nblk = sum(lv[0] * lv[1] for lv in pipe.level_grids())   # blocks one chain walks (all levels)
        out = {
            "metric": ("Compensate+BlockFPS 24->60 %s output fps (Super+Analyse+Compensate+BlockFPS end-to-end)" % args.config) if fpsconv else
                      ("MDegrain%d %s fps (Super+Analyse+Degrain end-to-end)" % (cfg[3], args.config)),
            "value": world * units * args.steps / dt, "unit": "fps", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16" if cfg[2] > 8 else "u8", "data": "synthetic",
            "config": {"workload": cfg[7], "frames_per_step_per_gpu": units, "input_frames_per_step_per_gpu": B + 1 if fpsconv else B, "chains_per_step_per_gpu": chains,
                       "sharding": "frame ranges, no collective", "batches_in_flight": len(pipes),
                       "rank0_output_frames": list(plan.out), "rank0_held_frames": list(plan.held)},
            "roofline": {"bound": "hbm", "kernel": "%s (the motion search; one launch = %d chains)" % (search_kernel_name(mv), chains), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "peak_measured": HBM_MEASURED_GBS, "frac_of_measured_peak": achieved / HBM_MEASURED_GBS,
                         "algorithmic_bytes_per_launch": bytes_chain * chains, "avg_launch_ms": avg_launch_ms, "avg_launch_event_ms": event_launch_ms,
                         "launch_time_basis": ("union of the launches' HIP-event intervals / launches: consecutive launches overlap (avg_launch_event_ms is the mean "
                                               "event-to-event duration of one launch, which counts the overlaps twice)") if overlapped else "mean HIP-event duration of a launch",
                         "search_share_of_step": avg_launch_ms * len(search_ms) / (dt * 1e3),
                         # the search is a serial chain per (frame, direction) -- its own yardstick is block steps per second
                         "blocks_per_chain": nblk, "chain_steps_per_s": chains * nblk / (avg_launch_ms * 1e-3)},
        }
        if len(pipes) > 1 and not fpsconv:
            # the launches of the timed steps ran beside the other slot's Super / Degrain kernels; the same launch with the GPU to itself (same clip, same result):
            alone_ms = pipes[(args.steps - 1) % len(pipes)].search_alone()
            alone_gbs = bytes_chain * chains / (alone_ms * 1e-3) / 1e9
            out["roofline"]["launch_alone"] = {
                "avg_launch_ms": alone_ms, "achieved": alone_gbs, "frac": alone_gbs / HBM_PEAK_GBS,
                "note": "one search launch after the timed region, nothing else on the GPU; `frac` above is the timed region's "
                        "(launches beside the neighbouring batches' Super / Degrain kernels and each other's tails)"}
        if len(pipes) > 1 and world == 1:
            # the same job with ONE batch in flight (the form rounds 1-4 were measured in): three untimed-then-timed steps of slot 0 alone, after the timed region
            pipes[0].step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                pipes[0].step()
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t1) / 3
            out["one_batch_in_flight"] = {"value": world * units / d1, "unit": "fps", "ms_per_step": d1 * 1e3, "steps": 3,
                                          "note": "--slots 1 equivalent, measured after the timed region: comparable with the headline of rounds 1-4"}
        if args.ingest and world == 1 and not fpsconv:
            sps, up_b, down_b = ingest_run(torch, pipe, max(2, min(args.steps, 4)), 1)
            out["ingest_inclusive"] = {"value": units / sps, "unit": "fps", "ms_per_step": sps * 1e3, "h2d_bytes_per_step": up_b, "d2h_bytes_per_step": down_b,
                                       "how": "source frames from pinned host memory (double-buffered on the device), output frames to pinned host memory, one copy stream, overlapped with compute"}
        if world == 1 and not (args.no_cpu and args.no_parity):
            # the CPU leg doubles as the parity check of the timed step: the oracle runs on frames of the SAME clip
            th = args.cpu_threads or min(os.cpu_count() or 1, 64)   # min(host cores, 64)
            F = (8 * th if fpsconv else th) if not args.no_cpu else 2
            base, parity = (oracle_leg_fps if fpsconv else oracle_leg)(mv, torch, cfg, pipes[(args.steps - 1) % len(pipes)] if args.steps else pipe, th, F)
            if not args.no_cpu:
                out["cpu_baseline"] = base
            out["parity_check"] = parity
            if not parity["identical"]:
                rc = 3
        if shard is not None:
            out["shard_check"] = shard
            if shard.get("shared_source_frames_identical") is False or any(b.get("identical") is False for b in shard.get("boundary_frames_vs_oracle", [])):
                rc = 3  # (a definite mismatch; an error of the side check itself is reported in the line only)
        traffic, traffic_note = None, "not measured (--no-traffic)" if args.no_traffic else "not measured at more than one rank"
        if world == 1 and not args.no_traffic:
            for pp in pipes:  # the profiled child needs the HBM this process holds
                pp.__dict__.clear()
            del pipe, pipes
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            traffic, traffic_note = measure_traffic(args, B)
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = traffic, traffic_note
        if "cpu_baseline" in out:
            # what the reference's own SIMD build would do with the same cores, relative to this scalar port (an estimate, not a measurement)
            out["cpu_baseline"]["reference_ratio_estimate"] = ("<= 2x this figure on 16-bit clips (the reference has no AVX2 kernels there: scalar C overlap-add, SSE2 SAD)" if cfg[2] > 8 else
                                                               "several x this figure on 8-bit clips (reference AVX2 SAD ~16x, overlap-add ~5-6x the scalar port per kernel)")
        if world == 1 and args.config == "cfg3" and not args.no_others and not args.no_cpu and not args.no_traffic and not args.batch:
            out["other_configs"] = other_configs()
            if not args.no_vs:
                out["vs_shell"] = vs_shell()
        print(json.dumps(out))
        if rc:
            sys.stderr.write("bench.py: the timed step's results differ from the oracle: %s\n" % out["parity_check"].get("mismatches"))
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
