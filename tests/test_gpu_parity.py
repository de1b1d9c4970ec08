"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.
Bit-exact: integer samples, motion vectors and SADs (no tolerance anywhere)."""
import ctypes as C

import numpy as np
import pytest

import pipeline as pl
import synth

pytestmark = pytest.mark.gpu


def _gpu_super(mv, sup, frames):
    dev = [mv.frame_to_device(f) for f in frames]
    out = sup.build(dev)
    import torch
    torch.cuda.synchronize()
    return dev, out


def _sup_to_numpy(mv, sup, frame):
    return [mv.plane_to_numpy(frame[p], sup.info.plane_width[p], sup.dtype) for p in range(sup.nplanes)]


SUPER_CASES = [
    # w, h, bits, sub, kwargs
    (128, 96, 8, (1, 1), {}),
    (128, 96, 16, (1, 1), {}),
    (160, 120, 8, (1, 1), dict(pel=1)),
    (160, 120, 10, (1, 1), dict(pel=4)),
    (136, 72, 8, (1, 1), dict(pel=4, sharp=0)),
    (200, 104, 8, (1, 1), dict(sharp=0)),
    (200, 104, 16, (1, 1), dict(sharp=1)),
    (144, 80, 8, (1, 1), dict(rfilter=0)),
    (144, 80, 8, (1, 1), dict(rfilter=1)),
    (144, 80, 16, (1, 1), dict(rfilter=3)),
    (144, 80, 8, (1, 1), dict(rfilter=4)),
    (130, 70, 8, (1, 1), dict(hpad=8, vpad=8)),       # odd-ish pyramid dims
    (134, 78, 8, (0, 0), dict(hpad=4, vpad=4)),       # 4:4:4
    (160, 96, 8, (1, 0), dict()),                     # 4:2:2
    (640, 360, 8, (1, 1), dict()),                    # BASELINE cfg1 size
    (128, 96, 8, (1, 1), dict(chroma=0)),
    (128, 96, 8, (1, 1), dict(levels=3)),
    # the rows-in-registers level-0 kernel (pel 2): widths that are no multiple of a thread's span, narrow frames (no fast columns at
    # all / a handful), paddings that keep or break the dword alignment it needs, every sharp mode, 4:4:4 / 4:2:2 / gray at 16 bits
    (138, 70, 16, (1, 1), {}),
    (138, 70, 16, (1, 1), dict(sharp=0)),
    (150, 66, 12, (1, 1), dict(sharp=1, hpad=2, vpad=2)),
    (40, 34, 16, (1, 1), dict(hpad=8, vpad=8)),
    (72, 40, 16, (1, 1), dict(hpad=0, vpad=0)),
    (150, 90, 16, (1, 1), dict(hpad=5, vpad=3)),      # chroma padding of 2 samples, luma of 5: alignment rule fails -> tile kernel
    (150, 90, 8, (1, 1), dict(hpad=6, vpad=6)),       # 8 bit: 6 % 4 != 0 -> tile kernel
    (150, 90, 8, (1, 1), dict(hpad=24, vpad=8, sharp=1)),
    (134, 78, 16, (0, 0), dict(hpad=4, vpad=4)),
    (160, 96, 16, (1, 0), dict()),
    (1000, 48, 16, (1, 1), dict()),                   # several waves per row
    (1000, 48, 8, (1, 1), dict()),
    (128, 96, 16, (1, 1), dict(chroma=0)),
]


@pytest.mark.parametrize("w,h,bits,sub,kw", SUPER_CASES)
def test_super_parity(oracle, mv, w, h, bits, sub, kw):
    frames = pl.moving_clip(w, h, bits, 2, seed=3, sub=sub)
    osup = oracle.Super(w, h, bits, subsampling=sub, **kw)
    gsup = mv.Super(w, h, bits, subsampling=sub, **kw)
    assert (gsup.info.super_width, gsup.info.super_height, gsup.info.levels) == (osup.s.superWidth, osup.s.superHeight, osup.s.levels)
    _, gout = _gpu_super(mv, gsup, frames)
    for f in range(2):
        of = osup.frame(frames[f])
        gf = _sup_to_numpy(mv, gsup, gout[f])
        bad = pl.defined_equal(osup, of, gf)
        assert not bad, "super frame %d differs in defined regions (plane, level, pelplane, count, y, x, oracle, gpu): %s" % (f, bad[:6])


@pytest.mark.parametrize("w,h,bits,sub,pel,padded,kw", [
    (128, 96, 8, (1, 1), 2, False, {}),
    (128, 96, 16, (1, 1), 2, False, {}),
    (136, 72, 8, (1, 1), 4, False, {}),
    (136, 72, 16, (1, 1), 4, False, dict(hpad=16, vpad=8)),
    (128, 96, 8, (1, 1), 2, True, {}),
    (136, 72, 16, (1, 1), 4, True, {}),
    (128, 96, 8, (0, 0), 2, False, {}),
    (128, 96, 8, (1, 1), 2, False, dict(chroma=0)),
])
def test_super_pelclip_parity(oracle, mv, w, h, bits, sub, pel, padded, kw):
    """mv.Super(pelclip=...) (MVSuper.c:229-256,:91-102; mvpRefineExt MVFrame.cpp:1529-1631): plain and pre-padded upsized clips,
    then a search on the resulting super clip (its sub-pel planes are whatever the user supplied)"""
    import torch
    frames = pl.moving_clip(w, h, bits, 2, seed=8, sub=sub, noise=3)
    osup = oracle.Super(w, h, bits, subsampling=sub, pel=pel, **kw)
    gsup = mv.Super(w, h, bits, subsampling=sub, pel=pel, **kw)
    pw, ph = ((w + 2 * osup.s.hpad) * pel, (h + 2 * osup.s.vpad) * pel) if padded else (w * pel, h * pel)
    assert osup.pelclip_mode(pw, ph) == gsup.pelclip_mode(pw, ph) == (2 if padded else 1)
    rng = np.random.default_rng(4)
    dt = np.uint8 if bits == 8 else np.uint16
    pelframes = [[rng.integers(0, 1 << bits, (ph >> (sub[1] if p else 0), pw >> (sub[0] if p else 0)), dtype=dt) for p in range(3)] for _ in frames]
    gsrc = [mv.frame_to_device(f) for f in frames]
    gpel = [mv.frame_to_device(f) for f in pelframes]
    gout = gsup.build(gsrc, pelclip=gpel, pelclip_size=(pw, ph))
    torch.cuda.synchronize()
    osf = []
    for f in range(2):
        of = osup.frame_pelclip(frames[f], pelframes[f])
        osf.append(of)
        bad = pl.defined_equal(osup, of, _sup_to_numpy(mv, gsup, gout[f]))
        assert not bad, "pelclip super frame %d differs (plane, level, pelplane, count, y, x, oracle, gpu): %s" % (f, bad[:6])
        assert pl.defined_equal(osup, of, osup.frame(frames[f])), "the pelclip must change the sub-pel planes"
    akw = dict(blksize=8, overlap=4, chroma=kw.get("chroma", 1))
    ob = oracle.Analyse(osup, isb=1, **akw).frame(osf[0], osf[1])
    gb = mv.Analyse(gsup, isb=1, **akw).run([(gout[0], gout[1])])[0]
    assert np.array_equal(gb.cpu().numpy(), ob)


def _behind(frame, p, offset, nbytes):
    """nbytes of device memory `offset` bytes behind plane p of a frame (its shadow data lives there)"""
    import torch
    t = frame[p]
    whole = torch.empty(0, dtype=torch.uint8, device=t.device).set_(t.untyped_storage())
    o = t.storage_offset() + offset
    return whole[o:o + nbytes].cpu().numpy()


@pytest.mark.parametrize("w,h,bits,sub,kw", [c for c in SUPER_CASES if c[2] > 8 and c[4].get("pel", 2) == 2])
def test_super_fused_shadow_planes(mv, w, h, bits, sub, kw):
    """mvx_super_frames_shadow (level-0 kernels write the shadow data themselves) against mvx_super_shadow_frames (derives it from
    the finished planes) on every shadow byte a search can read"""
    frames = pl.moving_clip(w, h, bits, 2, seed=4, sub=sub)
    gsup = mv.Super(w, h, bits, subsampling=sub, **kw)
    if not gsup.shadow:
        pytest.skip("no shadow planes for this format")
    _, gout = _gpu_super(mv, gsup, frames)
    i = gsup.info
    geo = []  # per plane: rows of the four level-0 planes, padded width
    for p in range(gsup.nplanes):
        xr, yr = (1, 1) if p == 0 else (i.xRatioUV, i.yRatioUV)
        geo.append((4 * (i.height // yr + 2 * (i.vpad // yr)), i.width // xr + 2 * (i.hpad // xr)))

    def grab(fr):
        size = [i.plane_height[p] * gsup.pitch[p] for p in range(gsup.nplanes)]
        luma = _behind(fr, 0, gsup.shadow_stride[0], size[0]).view(np.uint16).reshape(i.plane_height[0], -1).copy()
        uv = None
        if gsup.nplanes >= 3:
            uv = _behind(fr, 1, gsup.shadow_stride[1], 2 * size[1]).view(np.uint16).reshape(i.plane_height[1], -1).copy()
        return luma, uv
    fused = [grab(fr) for fr in gout]
    gsup._shadows(gout)
    import torch
    torch.cuda.synchronize()
    for f, fr in enumerate(gout):
        luma, uv = grab(fr)
        rows, pw = geo[0]
        assert np.array_equal(fused[f][0][:rows, :pw - 1], luma[:rows, :pw - 1])  # the last sample of a row has no right neighbour
        assert np.array_equal(fused[f][0][rows:], luma[rows:])
        if uv is not None and (i.modeYUV & 2):
            rows, pw = geo[1]
            assert np.array_equal(fused[f][1][:rows, :2 * pw], uv[:rows, :2 * pw])
            assert np.array_equal(fused[f][1][rows:], uv[rows:])


def test_super_pelclip_errors(mv):
    gsup = mv.Super(128, 96, 8, pel=2)
    with pytest.raises(mv.MvtoolsError, match="Super: pelclip's dimensions must be multiples of the input clip's dimensions."):
        gsup.pelclip_mode(128, 96)
    assert mv.Super(128, 96, 8, pel=1).pelclip_mode(77, 33) == 0  # pel 1: the pelclip is ignored (MVSuper.c:240)


def test_super_gray(oracle, mv):
    frames = [[p[0]] for p in pl.moving_clip(128, 96, 8, 1, seed=5)]
    osup = oracle.Super(128, 96, 8, gray=True)
    gsup = mv.Super(128, 96, 8, gray=True)
    _, gout = _gpu_super(mv, gsup, frames)
    assert not pl.defined_equal(osup, osup.frame(frames[0]), _sup_to_numpy(mv, gsup, gout[0]))


@pytest.fixture
def dbg(mv):
    """kernel-variant switches for one test (mvx_debug_option: selects among kernels that compute identical results); reset afterwards"""
    used = []

    def set_(name, value):
        mv.debug_option(name, value)
        used.append(name)
    yield set_
    for name in used:
        mv.debug_option(name, -1 if name in ("cpw_sync", "lds_min", "team") else 1 if name == "spec" else 0)


ANALYSE_CASES = [
    # w, h, bits, super kwargs, analyse kwargs
    (128, 96, 8, {}, dict(blksize=8, overlap=4)),
    (128, 96, 16, {}, dict(blksize=8, overlap=4)),
    (640, 360, 8, dict(pel=1), dict(blksize=8)),                         # BASELINE cfg1
    (320, 180, 8, {}, dict(blksize=8, overlap=4, search=4)),             # cfg2 shape, small
    (384, 224, 16, {}, dict(blksize=16, overlap=8)),                     # cfg3 shape, small
    (512, 288, 16, {}, dict(blksize=32, overlap=16)),                    # cfg5 shape, small
    (256, 144, 8, {}, dict(blksize=16, overlap=0)),
    (256, 144, 8, dict(pel=4), dict(blksize=8, overlap=2)),
    (256, 144, 8, {}, dict(blksize=4, overlap=2)),
    (256, 144, 8, {}, dict(blksize=8, blksizev=4, overlap=4, overlapv=2)),
    (256, 144, 8, {}, dict(blksize=16, blksizev=8, overlap=8, overlapv=4)),
    (256, 144, 8, {}, dict(blksize=8, chroma=0)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, meander=0)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, isb=1)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, truemotion=0)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=3, searchparam=2)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=3, searchparam=4, pelsearch=3)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=5, searchparam=8, pelsearch=8)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=0, searchparam=4, pelsearch=4)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=1, searchparam=3, pelsearch=3)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=2, searchparam=4, pelsearch=4)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=6, searchparam=5, pelsearch=5)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=7, searchparam=5, pelsearch=5)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search_coarse=4)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, trymany=1)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, badsad=300, badrange=8)),   # forces the UMH rescue on many blocks
    (256, 144, 8, {}, dict(blksize=8, overlap=4, badsad=300, badrange=-4)),  # exhaustive rescue
    (256, 144, 8, {}, dict(blksize=8, overlap=4, levels=2)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, global_=0, pglobal=20)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, plevel=2, lambda_=3000, lsad=800, pnew=20, pzero=70)),
    (200, 120, 8, dict(hpad=8, vpad=8), dict(blksize=8, overlap=2)),
    (320, 192, 8, {}, dict(blksize=8, overlap=4, _noise=14)),               # heavy noise: many bad blocks, ties
    (320, 192, 16, {}, dict(blksize=16, overlap=8, _noise=14)),
    (320, 192, 8, dict(pel=4), dict(blksize=8, overlap=4, _noise=10, badrange=6)),
    # SATD cost modes (PlaneOfBlocks.cpp:117-203); _lumaramp adds a brightness change so that dct 6-10 really mix SATD in
    (256, 144, 8, {}, dict(blksize=8, overlap=4, dct=5)),
    (256, 144, 16, {}, dict(blksize=16, overlap=8, dct=5)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, dct=6, _lumaramp=40)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, dct=7, _lumaramp=40)),
    (256, 144, 8, {}, dict(blksize=16, overlap=8, dct=8, _lumaramp=40)),
    (256, 144, 16, {}, dict(blksize=8, overlap=4, dct=9, _lumaramp=40)),
    (256, 144, 8, {}, dict(blksize=4, overlap=2, dct=10, _lumaramp=40)),
    (256, 144, 8, {}, dict(blksize=32, overlap=16, dct=5, chroma=0)),
    # every remaining block size of MVAnalyse.c:412 (generic kernel: run-time geometry)
    (512, 384, 8, {}, dict(blksize=64, overlap=32)),
    (512, 384, 16, {}, dict(blksize=64, blksizev=32, overlap=32, overlapv=16)),
    (640, 512, 8, {}, dict(blksize=128, overlap=64)),
    (640, 512, 16, {}, dict(blksize=128, blksizev=64, overlap=0)),
    (256, 144, 8, {}, dict(blksize=32, blksizev=16, overlap=16, overlapv=8)),
    (256, 144, 16, {}, dict(blksize=32, blksizev=16, overlap=0)),
    (256, 144, 8, {}, dict(blksize=16, blksizev=2, overlap=8, overlapv=0)),
    (256, 144, 16, {}, dict(blksize=16, blksizev=2, overlap=0, overlapv=0)),
    (256, 144, 8, {}, dict(blksize=32, overlap=16)),                       # 8-bit 32x32: generic kernel (the lean kernel has no 8-bit 32x32 build)
    (320, 192, 8, {}, dict(blksize=64, overlap=32, _noise=14, badsad=500)),  # rescue with big blocks
]


@pytest.mark.parametrize("w,h,bits,skw,akw", ANALYSE_CASES)
def test_analyse_parity(oracle, mv, w, h, bits, skw, akw):
    import torch
    akw = dict(akw)
    noise = akw.pop("_noise", 3)
    ramp = akw.pop("_lumaramp", 0)
    frames = pl.moving_clip(w, h, bits, 2, seed=11, noise=noise)
    if ramp:  # brightness change between the two frames (8-bit scale), stronger to the right: the luma-gated SATD modes fire
        f1 = frames[1]
        y = f1[0].astype(np.int64) + (np.linspace(0, ramp, w)[None, :] * (1 << (bits - 8))).astype(np.int64)
        f1[0][...] = np.clip(y, 0, (1 << bits) - 1).astype(f1[0].dtype)
    osup = oracle.Super(w, h, bits, **skw)
    gsup = mv.Super(w, h, bits, **skw)
    oan = oracle.Analyse(osup, **akw)
    gan = mv.Analyse(gsup, **akw)
    assert gan.blob_size == oan.blob_size
    for k, _ in oracle.AnalysisData._fields_:
        if k not in ("nMagicKey", "nVersion", "nCPUFlags"):  # never initialised / host dependent in the reference
            assert getattr(gan.ad, k) == getattr(oan.ad, k), k
    osf = [osup.frame(f) for f in frames]
    # feed the ORACLE's super frames to the GPU search so that this test isolates Analyse
    gsf = [gsup.from_host(sf) for sf in osf]
    blobs = gan.run([(gsf[0], gsf[1]), (gsf[1], gsf[0]), (gsf[0], None)])
    torch.cuda.synchronize()
    want = [oan.frame(osf[0], osf[1]), oan.frame(osf[1], osf[0]), oan.frame(osf[0], None)]
    for i in range(3):
        got = blobs[i].cpu().numpy()
        if not np.array_equal(got, want[i]):
            msg = []
            for lvl in range(oan.ad.nLvCount - 1, -1, -1):
                gx, gy, gs = pl.blob_vectors(got, oan.ad, lvl)
                wx, wy, ws = pl.blob_vectors(want[i], oan.ad, lvl)
                d = (gx != wx) | (gy != wy) | (gs != ws)
                if d.any():
                    ys, xs = np.nonzero(d)
                    y, x = int(ys[0]), int(xs[0])
                    msg.append("level %d: %d/%d blocks differ, first at (by=%d,bx=%d): gpu (%d,%d,%d) oracle (%d,%d,%d)" % (
                        lvl, int(d.sum()), d.size, y, x, gx[y, x], gy[y, x], gs[y, x], wx[y, x], wy[y, x], ws[y, x]))
            hdr = "header gpu %s oracle %s" % (got[:8].view(np.int32), want[i][:8].view(np.int32))
            pytest.fail("job %d blob differs: %s | %s" % (i, hdr, " ; ".join(msg[:4])))


def _pipeline(oracle, mv, w, h, bits, radius, skw, akw, nframes=None, seed=21):
    import torch
    nframes = nframes or (2 * radius + 1)
    frames = pl.moving_clip(w, h, bits, nframes, seed=seed, noise=3)
    osup = oracle.Super(w, h, bits, **skw)
    gsup = mv.Super(w, h, bits, **skw)
    osf = [osup.frame(f) for f in frames]
    gsrc = [mv.frame_to_device(f) for f in frames]
    gsf = gsup.build(gsrc)
    return frames, osup, gsup, osf, gsrc, gsf


DEGRAIN_CASES = [
    (128, 96, 8, 1, {}, dict(blksize=8, overlap=4), {}),
    (128, 96, 16, 1, {}, dict(blksize=8, overlap=4), {}),
    (192, 112, 8, 2, {}, dict(blksize=16, overlap=8), {}),
    (192, 112, 16, 3, {}, dict(blksize=16, overlap=8), {}),           # cfg3 shape
    (256, 160, 16, 6, {}, dict(blksize=32, overlap=16), {}),          # cfg5 shape (tr=6)
    (196, 116, 16, 1, {}, dict(blksize=16, overlap=8), {}),           # width not a multiple of the cell: partial cell rows
    (196, 116, 8, 2, {}, dict(blksize=8, overlap=4), {}),
    (200, 120, 8, 1, {}, dict(blksize=8, overlap=0), {}),             # no overlap + uncovered strips
    (200, 120, 8, 1, dict(pel=1), dict(blksize=8, overlap=2), {}),
    (200, 120, 8, 1, dict(pel=4), dict(blksize=8, overlap=4), {}),
    (128, 96, 8, 1, {}, dict(blksize=8, overlap=4), dict(limit=3, limitc=5)),
    (128, 96, 8, 1, {}, dict(blksize=8, overlap=4), dict(plane=0)),
    (128, 96, 8, 1, {}, dict(blksize=8, overlap=4), dict(plane=3, thsadc=150)),
    (128, 96, 8, 1, {}, dict(blksize=8, overlap=4), dict(thsad=100, thscd1=200, thscd2=60)),
    (128, 96, 8, 1, {}, dict(blksize=8, overlap=4), dict(thscd1=20, thscd2=10)),  # scene change: refs unusable
    (192, 112, 8, 4, {}, dict(blksize=16, overlap=8), {}),            # Degrain4 / Degrain5 (MVDegrains.cpp:432-469)
    (192, 112, 16, 5, {}, dict(blksize=8, overlap=4), {}),
    (512, 384, 8, 1, {}, dict(blksize=64, overlap=32), {}),           # big blocks: the per-sample gather
    (512, 384, 16, 2, {}, dict(blksize=64, blksizev=32, overlap=32, overlapv=16), {}),
    (640, 512, 8, 1, {}, dict(blksize=128, overlap=64), {}),
    (256, 144, 16, 1, {}, dict(blksize=32, blksizev=16, overlap=0), {}),
    (256, 144, 8, 1, {}, dict(blksize=16, blksizev=2, overlap=8, overlapv=0), {}),
    # blocks side by side through the cell kernel (r5) beyond the two plain cases above: several v_dot2 pairs on 16 bit with 16x16 blocks
    # (chroma cell W = 8), the limits, a luma-only run, unusable references (the `safe` pointer) -- ADVICE r5
    (192, 112, 16, 3, {}, dict(blksize=16, overlap=0), {}),
    (128, 96, 8, 1, {}, dict(blksize=8, overlap=0), dict(limit=3, limitc=5)),
    (128, 96, 16, 2, {}, dict(blksize=16, overlap=0), dict(limit=2, limitc=4)),
    (128, 96, 16, 1, {}, dict(blksize=16, overlap=0), dict(plane=0)),
    (128, 96, 8, 1, {}, dict(blksize=16, overlap=0), dict(thscd1=20, thscd2=10)),
    # r6: the plan tile of the cell kernel (six or more references: Degrain3-6) on frames several workgroup tiles wide (a tile is 32 cells x 8 rows: the first block column
    # of a tile is the one that starts LEFT of its first cell), ragged at the right and bottom edges, 8- and 16-bit, with the limits, one plane, unusable references
    (544, 168, 8, 3, {}, dict(blksize=16, overlap=8), {}),                     # 68 x 21 cells of 8: three tile columns, the last one partial
    (548, 172, 16, 3, {}, dict(blksize=8, overlap=4), dict(limit=3, limitc=5)),  # 137 x 43 cells of 4 (chroma: of 2), width and height not multiples of the cell
    (400, 200, 16, 6, {}, dict(blksize=16, overlap=8), dict(thscd1=20, thscd2=10)),  # twelve references, all unusable: every load goes to the `safe` plane
    (400, 200, 8, 4, {}, dict(blksize=16, overlap=8), dict(plane=0)),
    (576, 160, 16, 3, {}, dict(blksize=32, blksizev=16, overlap=16, overlapv=8), {}),  # 16-sample cells, block rows of 8
]


@pytest.mark.parametrize("w,h,bits,radius,skw,akw,dkw", DEGRAIN_CASES)
def test_degrain_parity(oracle, mv, w, h, bits, radius, skw, akw, dkw):
    import torch
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, radius, skw, akw)
    n = len(frames)
    mid = radius  # output frame; also test the clip edge (frame 0: forward refs missing)
    for target in (mid, 0):
        oblobs, gjobs, refs_o, refs_g = [], [], [], []
        for d in range(1, radius + 1):
            for isb in (1, 0):
                oan = oracle.Analyse(osup, isb=isb, delta=d, **akw)
                gan = mv.Analyse(gsup, isb=isb, delta=d, **akw)
                nref = target + (d if isb else -d)
                ok = 0 <= nref < n
                oblobs.append(oan.frame(osf[target], osf[nref] if ok else None))
                gjobs.append((gan, (gsf[target], gsf[nref] if ok else None)))
                refs_o.append(osf[nref] if ok else None)
                refs_g.append(gsf[nref] if ok else None)
        gblobs = [gan.run([job])[0] for gan, job in gjobs]
        for a, b in zip(gblobs, oblobs):
            assert np.array_equal(a.cpu().numpy(), b), "vectors differ (Super+Analyse through the GPU)"
        odg = oracle.Degrain(radius, osup, oan.ad, **dkw)
        gdg = mv.Degrain(radius, gsup, gan.ad, [p.stride(0) for p in gsrc[0]], **dkw)
        want = odg.frame(frames[target], refs_o, oblobs)
        got = gdg.run([(gsrc[target], refs_g, gblobs)])[0]
        torch.cuda.synchronize()
        for p in range(3):
            g = mv.plane_to_numpy(got[p], want[p].shape[1], want[p].dtype)
            if not np.array_equal(g, want[p]):
                ys, xs = np.nonzero(g != want[p])
                pytest.fail("degrain target %d plane %d: %d samples differ, first (y=%d,x=%d) gpu %d oracle %d" % (
                    target, p, len(ys), ys[0], xs[0], g[ys[0], xs[0]], want[p][ys[0], xs[0]]))


COMP_CASES = [
    (128, 96, 8, {}, dict(blksize=8, overlap=4), {}),
    (192, 112, 16, {}, dict(blksize=16, overlap=8), {}),
    (200, 120, 8, {}, dict(blksize=8, overlap=0), {}),
    (200, 120, 8, {}, dict(blksize=8, overlap=0), dict(scbehavior=0)),
    (128, 96, 8, {}, dict(blksize=8, overlap=4), dict(thsad=60)),
    (128, 96, 8, {}, dict(blksize=8, overlap=4), dict(time=40.0)),
    (128, 96, 8, {}, dict(blksize=8, overlap=4), dict(thscd1=20, thscd2=10)),
]


@pytest.mark.parametrize("w,h,bits,skw,akw,ckw", COMP_CASES)
def test_compensate_parity(oracle, mv, w, h, bits, skw, akw, ckw):
    import torch
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, skw, akw, nframes=2)
    oan = oracle.Analyse(osup, isb=1, **akw)
    gan = mv.Analyse(gsup, isb=1, **akw)
    for (src, ref) in ((0, 1), (1, None)):
        ob = oan.frame(osf[src], osf[ref] if ref is not None else None)
        gb = gan.run([(gsf[src], gsf[ref] if ref is not None else None)])[0]
        assert np.array_equal(gb.cpu().numpy(), ob)
        oc = oracle.Compensate(osup, oan.ad, **ckw)
        gc = mv.Compensate(gsup, gan.ad, **ckw)
        want = oc.frame(osf[src], osf[ref] if ref is not None else None, ob)
        got = gc.run([(gsf[src], gsf[ref] if ref is not None else None, gb)])[0]
        torch.cuda.synchronize()
        for p in range(3):
            g = mv.plane_to_numpy(got[p], want[p].shape[1], want[p].dtype)
            assert np.array_equal(g, want[p]), "compensate plane %d differs (%d samples)" % (p, int((g != want[p]).sum()))


@pytest.mark.parametrize("w,h,bits,pel,akw,shift", [
    (128, 96, 8, 2, dict(blksize=8, overlap=4), 1),
    (128, 96, 8, 2, dict(blksize=8, overlap=0), -1),
    (192, 112, 16, 4, dict(blksize=16, overlap=8), 2),
    (192, 112, 16, 4, dict(blksize=16, overlap=8), -2),
    (128, 96, 8, 2, dict(blksize=8, overlap=4, thsad=60), 1),
])
def test_fields_shift_parity(oracle, mv, w, h, bits, pel, akw, shift):
    """fields=True (MVAnalyse.c:172-176, MVCompensate.c:188-225): the +-pel/2 vertical shift between fields of opposite
    parity, in the search (zero / global predictors) and in Compensate's block fetch (both the vector and the fallback)"""
    import torch
    ckw = {}
    akw = dict(akw)
    if "thsad" in akw:
        ckw["thsad"] = akw.pop("thsad")
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, dict(pel=pel), akw, nframes=2)
    oan = oracle.Analyse(osup, isb=1, fields=1, **akw)
    gan = mv.Analyse(gsup, isb=1, fields=1, **akw)
    ob = oan.frame(osf[0], osf[1], field_shift=shift)
    gb = gan.run([(gsf[0], gsf[1])], field_shift=shift)[0]
    assert np.array_equal(gb.cpu().numpy(), ob)
    oc = oracle.Compensate(osup, oan.ad, **ckw)
    gc = mv.Compensate(gsup, gan.ad, fields=1, **ckw)
    want = oc.frame(osf[0], osf[1], ob, field_shift=shift)
    got = gc.run([(gsf[0], gsf[1], gb, shift)])[0]
    torch.cuda.synchronize()
    for p in range(3):
        g = mv.plane_to_numpy(got[p], want[p].shape[1], want[p].dtype)
        assert np.array_equal(g, want[p]), "compensate plane %d differs (%d samples)" % (p, int((g != want[p]).sum()))
    assert not all(np.array_equal(a, b) for a, b in zip(want, oc.frame(osf[0], osf[1], ob)))


@pytest.mark.parametrize("bits,akw", [(8, dict(blksize=8, overlap=4)), (16, dict(blksize=16, overlap=8)), (16, dict(blksize=32, overlap=16))])
def test_analyse_one_chain_per_workgroup(oracle, mv, dbg, bits, akw):
    """MVX_CPW=1 keeps the one-chain-per-workgroup builds of the specialised kernels reachable (A/B timing uses them): same vectors"""
    dbg("cpw1", 1)
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, 256, 160, bits, 1, {}, akw, nframes=2, seed=9)
    ob = oracle.Analyse(osup, isb=1, **akw).frame(osf[0], osf[1])
    gb = mv.Analyse(gsup, isb=1, **akw).run([(gsf[0], gsf[1])])[0]
    assert np.array_equal(gb.cpu().numpy(), ob)


@pytest.mark.parametrize("bits,akw", [(8, dict(blksize=8, overlap=4)), (16, dict(blksize=16, overlap=8)), (16, dict(blksize=8, overlap=4)),
                                      (8, dict(blksize=16, overlap=8)), (16, dict(blksize=32, overlap=16))])
@pytest.mark.parametrize("kernel", ["lean", "general"])
def test_analyse_two_chains_per_simd(oracle, mv, dbg, kernel, bits, akw):
    """a launch with more chains than the device has SIMDs takes the 256-register builds (two chains per SIMD; 16-bit: eight
    chains per workgroup, job table sorted by reference frame): every one of its results must still be the oracle's, whatever
    the order the chains were given in"""
    import torch
    if kernel == "general":
        dbg("general", 1)
    w, h = 128, 96
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, {}, akw, nframes=3)
    oan = oracle.Analyse(osup, isb=1, **akw)
    gan = mv.Analyse(gsup, isb=1, **akw)
    want = [oan.frame(osf[0], osf[1]), oan.frame(osf[1], osf[2]), oan.frame(osf[2], None), oan.frame(osf[0], osf[2])]
    pairs = [(gsf[0], gsf[1]), (gsf[1], gsf[2]), (gsf[2], None), (gsf[0], gsf[2])]
    nsimd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    njobs = nsimd + 61
    got = gan.run([pairs[(i * 7) % 4] for i in range(njobs)])
    torch.cuda.synchronize()
    got = torch.stack(list(got)).cpu().numpy()
    for i in range(njobs):
        assert np.array_equal(got[i], want[(i * 7) % 4]), "job %d differs" % i


@pytest.mark.parametrize("w,h,bits,skw,akw", [
    (128, 96, 8, {}, dict(blksize=8, overlap=4)),
    (384, 224, 16, {}, dict(blksize=16, overlap=8)),
    (512, 288, 16, {}, dict(blksize=32, overlap=16)),
    (256, 144, 8, {}, dict(blksize=16, overlap=0)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, search=3, searchparam=2)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, badsad=300, badrange=8)),
    (256, 144, 8, {}, dict(blksize=8, overlap=4, badsad=300, badrange=-4)),
    (320, 192, 16, {}, dict(blksize=16, overlap=8, _noise=14)),
    (256, 144, 8, dict(pel=4), dict(blksize=8, overlap=2)),
    (256, 144, 8, dict(pel=1), dict(blksize=8, chroma=0)),
])
@pytest.mark.parametrize("variant", ["general", "plain-layout", "serial", "spec-off", "spec-everywhere"])
def test_analyse_default_search_other_kernels(oracle, mv, dbg, variant, w, h, bits, skw, akw):
    """The default search normally runs in the speculative kernel (mvx_analyse_spec.h) on super frames that carry shadow copies.
    The same cases through the general kernel ("general" = 1), through the speculative kernel on the plain layout (no shadow
    copies: unaligned loads), through the serial lean kernel (mvx_analyse_fast.h, "spec" = 0) and through the speculative kernel's
    code with speculation switched off ("spec" = 2: every block searched live) or forced for shapes it is not the default for ("spec" = 5)
    must give the same blobs."""
    akw = dict(akw)
    noise = akw.pop("_noise", 3)
    if variant == "general":
        dbg("general", 1)
    if variant == "serial":
        dbg("spec", 0)
    if variant == "spec-off":
        dbg("spec", 2)
    if variant == "spec-everywhere":
        dbg("spec", 5)
    frames = pl.moving_clip(w, h, bits, 3, seed=13, noise=noise)
    osup = oracle.Super(w, h, bits, **skw)
    gsup = mv.Super(w, h, bits, shadow=(variant != "plain-layout"), **skw)
    osf = [osup.frame(f) for f in frames]
    gsf = gsup.build([mv.frame_to_device(f) for f in frames])
    for isb in (1, 0):
        oan = oracle.Analyse(osup, isb=isb, **akw)
        gan = mv.Analyse(gsup, isb=isb, **akw)
        ref = 2 if isb else 0
        assert np.array_equal(gan.run([(gsf[1], gsf[ref])])[0].cpu().numpy(), oan.frame(osf[1], osf[ref]))


WINDOW_CASES = [
    (384, 224, {}, dict(blksize=16, overlap=8)),                                  # cfg3 shape
    (320, 192, {}, dict(blksize=16, overlap=8, _noise=14)),                       # many bad blocks: the rescue (global path) between window blocks
    (320, 192, {}, dict(blksize=16, overlap=8, _noise=14, badsad=400, badrange=-3)),
    (256, 144, {}, dict(blksize=16, overlap=0)),
    (256, 144, {}, dict(blksize=16, overlap=8, chroma=0)),
    (256, 144, dict(pel=1), dict(blksize=16, overlap=8)),
    (256, 144, {}, dict(blksize=16, overlap=4, search=3, searchparam=2, pelsearch=2)),   # exhaustive radius 2 at the finest level too
    (256, 144, {}, dict(blksize=16, overlap=8, pelsearch=1)),                        # Hex2 with range 1: the square only
    (256, 144, {}, dict(blksize=16, overlap=8, global_=0, pglobal=30, pzero=90)),
    (256, 144, {}, dict(blksize=16, overlap=8, meander=0, levels=2)),
    (200, 120, dict(hpad=8, vpad=8), dict(blksize=16, overlap=8)),               # small padding: windows clamp at the plane edges
    (1000, 64, {}, dict(blksize=16, overlap=8)),                                  # several 64-block groups per row
]


@pytest.mark.parametrize("w,h,skw,akw", WINDOW_CASES + [
    (1000, 96, {}, dict(blksize=16, overlap=8, meander=0)),                       # several groups per row, always left to right
    (1000, 96, {}, dict(blksize=16, overlap=8, chroma=0)),                        # r5: a luma-only search through the row passes (no UV rows), several groups per row
    (330, 192, {}, dict(blksize=16, overlap=0, chroma=0, _noise=14)),
    (1000, 96, {}, dict(blksize=16, overlap=0)),                                  # r5: blocks side by side (windows of four), several groups per row
    (330, 192, {}, dict(blksize=16, overlap=0, _noise=14, badsad=400, badrange=6)),  # ... a ragged last window, rescues
    (640, 360, {}, dict(blksize=16, overlap=8, _noise=0)),                        # a clean clip: long verified runs
    (320, 192, {}, dict(blksize=16, overlap=8, _noise=14, badsad=400, badrange=6)),  # UMH rescue between verified runs
    (520, 96, dict(hpad=4, vpad=4), dict(blksize=16, overlap=8, pglobal=20)),     # tiny padding: the global predictor is clipped block by block
])
@pytest.mark.parametrize("mode", ["default", "everywhere", "no-runs", "team"])
def test_analyse_speculative_kernel(oracle, mv, dbg, mode, w, h, skw, akw):
    _speculative_case(oracle, mv, dbg, mode, w, h, 16, skw, akw)


# the same kernel on 8-bit clips: row passes over windows of up to fifteen 8x8 blocks overlapping by four (a half block = one dword)
ROWS8_CASES = [
    (384, 224, {}, dict(blksize=8, overlap=4)),                                   # cfg2 shape
    (320, 192, {}, dict(blksize=8, overlap=4, _noise=14)),                        # most hypotheses fail
    (320, 192, {}, dict(blksize=8, overlap=4, _noise=14, badsad=300, badrange=-3)),
    (256, 144, {}, dict(blksize=8, overlap=0)),                                   # blocks side by side: windows of eight
    (640, 360, dict(pel=1), dict(blksize=8, overlap=0)),                          # cfg1 shape
    (320, 192, {}, dict(blksize=8, overlap=0, _noise=14, search=3, searchparam=2, pelsearch=2)),
    (136, 96, dict(hpad=4, vpad=4), dict(blksize=8, overlap=0, pglobal=20)),
    (600, 64, {}, dict(blksize=8, overlap=0, meander=0)),                         # 75 blocks per row: a last group of eleven
    (256, 144, {}, dict(blksize=8, overlap=2)),                                   # another overlap: no row passes (one block at a time / the lean kernel)
    (256, 144, {}, dict(blksize=8, overlap=4, chroma=0)),
    (256, 144, dict(pel=1), dict(blksize=8, overlap=4)),
    (256, 144, {}, dict(blksize=8, overlap=4, search=3, searchparam=2, pelsearch=2)),  # exhaustive radius 2 at the finest level too
    (256, 144, {}, dict(blksize=8, overlap=4, pelsearch=1)),
    (256, 144, {}, dict(blksize=8, overlap=4, global_=0, pglobal=30, pzero=90)),
    (256, 144, {}, dict(blksize=8, overlap=4, meander=0, levels=2)),
    (200, 120, dict(hpad=8, vpad=8), dict(blksize=8, overlap=4)),
    (136, 96, dict(hpad=4, vpad=4), dict(blksize=8, overlap=4, pglobal=20)),      # tiny padding: candidates reach the very end of a row
    (1000, 64, {}, dict(blksize=8, overlap=4)),                                   # several 64-block groups per row; a short last group
    (532, 64, {}, dict(blksize=8, overlap=4, meander=0)),                         # 132 blocks per row: a last group of four
    (524, 64, {}, dict(blksize=8, overlap=4)),                                    # 130 blocks per row: a last group of two (one block at a time)
    (640, 360, {}, dict(blksize=8, overlap=4, _noise=0)),                         # a clean clip: long verified runs
]


@pytest.mark.parametrize("w,h,skw,akw", ROWS8_CASES)
@pytest.mark.parametrize("mode", ["default", "everywhere", "team"])
def test_analyse_speculative_kernel_8bit(oracle, mv, dbg, mode, w, h, skw, akw):
    _speculative_case(oracle, mv, dbg, mode, w, h, 8, skw, akw)


@pytest.mark.parametrize("w,h,skw,akw", WINDOW_CASES + [
    (1000, 96, {}, dict(blksize=16, overlap=8, meander=0)),
    (640, 360, {}, dict(blksize=16, overlap=8, _noise=0)),
    (520, 96, dict(hpad=4, vpad=4), dict(blksize=16, overlap=8, pglobal=20)),     # tiny padding: candidates reach the very end of a row (8-byte loads, no over-read)
    (1920, 64, {}, dict(blksize=16, overlap=8)),                                  # a 1080p row of blocks: 239 per row
    (1000, 96, {}, dict(blksize=16, overlap=0)),                                  # blocks side by side
    (330, 192, {}, dict(blksize=16, overlap=0, _noise=14, pelsearch=1)),
    (1000, 96, {}, dict(blksize=16, overlap=8, chroma=0, _noise=14)),             # luma-only
])
@pytest.mark.parametrize("mode", ["default", "everywhere", "team"])
def test_analyse_speculative_kernel_8bit_16x16(oracle, mv, dbg, mode, w, h, skw, akw):
    """r5: 8-bit clips, 16x16 blocks overlapping by 8 (the common HD setting) through the row passes of the 16-bit form with 8-byte columns."""
    _speculative_case(oracle, mv, dbg, mode, w, h, 8, skw, akw)


@pytest.mark.parametrize("nw", [2, 3, 5, 8])
@pytest.mark.parametrize("w,h,bits,skw,akw", [
    (1000, 96, 16, {}, dict(blksize=16, overlap=8)),                              # four groups per row, meander: the turn-around waits for the token
    (1000, 96, 16, {}, dict(blksize=16, overlap=8, meander=0, _noise=14)),        # always left to right; most hypotheses fail
    (1000, 64, 8, {}, dict(blksize=8, overlap=4)),
    (320, 192, 16, {}, dict(blksize=16, overlap=8, _noise=14, badsad=400, badrange=6)),  # rescues: badcount travels with the token
    (512, 288, 16, {}, dict(blksize=32, overlap=16)),                             # one group per row (cfg5 shape)
])
def test_analyse_team_sizes(oracle, mv, dbg, nw, w, h, bits, skw, akw):

    """The team form of the speculative kernel (mvx_analyse_spec.h, TEAM: the nw waves of a workgroup walk one chain, the token passes from group to
    group) with 2, 3, 5 and 8 waves per chain: same blobs as the oracle whatever the number of waves."""
    _speculative_case(oracle, mv, dbg, "team%d" % nw, w, h, bits, skw, akw)


@pytest.mark.parametrize("w,h,bits,skw,akw", [
    (384, 224, 16, {}, dict(blksize=16, overlap=8)),      # cfg3 shape: a launch of two chains leaves the GPU empty -> teams of four
    (384, 224, 8, {}, dict(blksize=8, overlap=4)),        # cfg2 shape
    (512, 288, 16, {}, dict(blksize=32, overlap=16)),     # no row passes: still teams (the speculative kernel one block at a time beats the serial walk of a lone chain)
    (256, 144, 8, {}, dict(blksize=16, overlap=8)),       # the common HD shape, likewise
])
def test_analyse_team_is_the_librarys_choice_for_small_launches(oracle, mv, dbg, w, h, bits, skw, akw):
    """With nothing forced, a launch that would leave wave slots empty at one wave per chain runs as teams (mvx_analyse.hip: mvx_team_default)."""
    _speculative_case(oracle, mv, dbg, "auto", w, h, bits, skw, akw)
    info = (C.c_int * 5)()
    mv.lib().mvx_debug_last_launch(info)
    assert info[4] == 3 and info[1] == 4, list(info)


def _speculative_case(oracle, mv, dbg, mode, w, h, bits, skw, akw):
    """The speculative kernel of the default search (mvx_analyse_spec.h): groups of 32 blocks evaluated ahead of the serial walk under
    the hypothesis left == up (row passes over windows of seven blocks, strip or block form), verified block by block, everything else
    searched live.  Same blobs as the oracle -- forward, backward, with a field shift, with a missing reference; noisy clips (most
    hypotheses fail), rescues, every refinement shape.  "default": the library's own choice (the speculative kernel where its row passes
    apply, the serial lean kernel elsewhere); "everywhere": forced for every shape it can run ("spec" = 5: the one-block-at-a-time
    passes); "no-runs": forced, without row passes ("spec" = 3)."""
    akw = dict(akw)
    noise = akw.pop("_noise", 3)
    if mode == "everywhere":
        dbg("spec", 5)
    if mode == "no-runs":
        dbg("spec", 3)
    if mode.startswith("team"):  # the team form, forced for every shape the speculative kernel can run
        dbg("spec", 5)
        dbg("team", int(mode[4:] or 4))
    elif mode != "auto":         # one wave per chain (the library itself picks the team form for launches this small: "auto")
        dbg("team", 0)
    blk, ov = akw.get("blksize", 8), akw.get("overlap", 0)  # row passes: 16-bit 16x16 overlapping by half; 8-bit 8x8 overlapping by half or not at all
    rows_apply = (blk == 16 and ov in (8, 0)) or (akw.get("chroma", 1) != 0 and (bits, blk) == (8, 8) and ov in (4, 0))  # (16x16: luma-only searches too, r5)
    frames = pl.moving_clip(w, h, bits, 3, seed=17, noise=noise, motion=(5, -2))
    osup = oracle.Super(w, h, bits, **skw)
    gsup = mv.Super(w, h, bits, **skw)
    osf = [osup.frame(f) for f in frames]
    gsf = gsup.build([mv.frame_to_device(f) for f in frames])
    info = (C.c_int * 5)()
    for isb in (1, 0):
        oan = oracle.Analyse(osup, isb=isb, **akw)
        gan = mv.Analyse(gsup, isb=isb, **akw)
        ref = 2 if isb else 0
        got = gan.run([(gsf[1], gsf[ref]), (gsf[1], None)])
        mv.lib().mvx_debug_last_launch(info)
        assert info[4] == (3 if (mode.startswith("team") or mode == "auto") else 2 if (mode != "default" or rows_apply) else 0), "not the kernel this case is meant to cover (%s)" % list(info)
        assert np.array_equal(got[0].cpu().numpy(), oan.frame(osf[1], osf[ref]))
        assert np.array_equal(got[1].cpu().numpy(), oan.frame(osf[1], None))
        if skw.get("pel", 2) == 2:
            for fs in (1, -1):  # fields: the zero candidate's luma is shifted, its chroma is not (PlaneOfBlocks.cpp:836-839)
                assert np.array_equal(gan.run([(gsf[1], gsf[ref])], field_shift=fs)[0].cpu().numpy(), oan.frame(osf[1], osf[ref], field_shift=fs))


@pytest.mark.parametrize("bits,akw,per_simd", [(8, dict(blksize=8, overlap=4), 3), (8, dict(blksize=8, overlap=4), 4), (16, dict(blksize=16, overlap=8), 3),
                                               (16, dict(blksize=16, overlap=8), 4), (8, dict(blksize=16, overlap=8), 4), (16, dict(blksize=32, overlap=16), 3),
                                               (16, dict(blksize=8, overlap=4), 4)])
def test_analyse_many_chains_per_simd(oracle, mv, dbg, bits, akw, per_simd):
    """launches with more than two (three) chains per SIMD take the lean kernel's 168- (128-) register builds, workgroups of twelve
    (sixteen) chains: every result must still be the oracle's, whatever the order the chains were given in.  ("spec" = 0: the shapes
    the speculative kernel's row passes cover would otherwise run there, at two per SIMD.)"""
    import torch
    dbg("spec", 0)
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, 96, 64, bits, 1, {}, akw, nframes=3)
    oan = oracle.Analyse(osup, isb=0, **akw)
    gan = mv.Analyse(gsup, isb=0, **akw)
    want = [oan.frame(osf[1], osf[0]), oan.frame(osf[2], osf[1]), oan.frame(osf[0], None), oan.frame(osf[2], osf[0])]
    pairs = [(gsf[1], gsf[0]), (gsf[2], gsf[1]), (gsf[0], None), (gsf[2], gsf[0])]
    njobs = (per_simd - 1) * 4 * torch.cuda.get_device_properties(0).multi_processor_count + 53
    got = gan.run([pairs[(i * 5) % 4] for i in range(njobs)])
    torch.cuda.synchronize()
    got = torch.stack(list(got)).cpu().numpy()
    for i in range(njobs):
        assert np.array_equal(got[i], want[(i * 5) % 4]), "job %d differs" % i


def test_analyse_three_chains_per_simd(oracle, mv, dbg):
    """more than two chains per SIMD: the general 8-bit 8x8 kernel's 168-register build (three chains per SIMD)"""
    dbg("general", 1)
    import torch
    akw = dict(blksize=8, overlap=4)
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, 96, 64, 8, 1, {}, akw, nframes=3)
    oan = oracle.Analyse(osup, isb=0, **akw)
    gan = mv.Analyse(gsup, isb=0, **akw)
    want = [oan.frame(osf[1], osf[0]), oan.frame(osf[2], osf[1]), oan.frame(osf[0], None)]
    pairs = [(gsf[1], gsf[0]), (gsf[2], gsf[1]), (gsf[0], None)]
    njobs = 8 * torch.cuda.get_device_properties(0).multi_processor_count + 37
    got = gan.run([pairs[(i * 5) % 3] for i in range(njobs)])
    torch.cuda.synchronize()
    got = torch.stack(list(got)).cpu().numpy()
    for i in range(njobs):
        assert np.array_equal(got[i], want[(i * 5) % 3]), "job %d differs" % i


def test_analyse_wide_frame_many_chains(oracle, mv):
    """8K-wide 16-bit frames: the row buffer of a chain is ~20 KiB, so eight chains no longer fit a CU's LDS and a launch with more
    chains than SIMDs falls back to four per workgroup (mvx_analyse_frames)"""
    import torch
    w, h, bits = 7680, 80, 16
    akw = dict(blksize=16, overlap=8)
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, {}, akw, nframes=2, seed=12)
    oan = oracle.Analyse(osup, isb=1, **akw)
    gan = mv.Analyse(gsup, isb=1, **akw)
    want = [oan.frame(osf[0], osf[1]), oan.frame(osf[1], osf[0])]
    pairs = [(gsf[0], gsf[1]), (gsf[1], gsf[0])]
    njobs = 4 * torch.cuda.get_device_properties(0).multi_processor_count + 8
    got = gan.run([pairs[i % 2] for i in range(njobs)])
    torch.cuda.synchronize()
    got = torch.stack(list(got)).cpu().numpy()
    for i in range(njobs):
        assert np.array_equal(got[i], want[i % 2]), "job %d differs" % i


def _golden_cases():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["name"])
def test_gpu_against_golden_fixtures(oracle, mv, case):
    """the HIP path against the committed golden vectors (tests/golden/golden.json) -- no oracle computation involved,
    mvoracle is only used for its FNV-1a hash helper and the list of defined super-frame rectangles"""
    import torch
    c = case["params"]
    w, h, bits, radius = c["w"], c["h"], c["bits"], c["radius"]
    frames = pl.moving_clip(w, h, bits, 2 * radius + 1, seed=99, noise=3)
    gsup = mv.Super(w, h, bits, **c["skw"])
    gsrc = [mv.frame_to_device(f) for f in frames]
    gsf = gsup.build(gsrc)
    osup = oracle.Super(w, h, bits, **c["skw"])  # geometry only
    sf_np = _sup_to_numpy(mv, gsup, gsf[radius])
    hs = [oracle.fnv1a(np.ascontiguousarray(sf_np[p][y0:y0 + hh, x0:x0 + ww])) for (p, lv, k, y0, x0, hh, ww) in osup.defined_regions()]
    assert "%08x" % oracle.fnv1a(np.array(hs, dtype=np.uint32)) == case["expect"]["super_regions_fnv"]
    blobs, refs = [], []
    for d in range(1, radius + 1):
        for isb in (1, 0):
            gan = mv.Analyse(gsup, isb=isb, delta=d, **c["akw"])
            nref = radius + (d if isb else -d)
            blobs.append(gan.run([(gsf[radius], gsf[nref])])[0])
            refs.append(gsf[nref])
    assert ["%08x" % oracle.fnv1a(b.cpu().numpy()) for b in blobs] == case["expect"]["blobs_fnv"]
    out = mv.Degrain(radius, gsup, gan.ad, [p.stride(0) for p in gsrc[0]]).run([(gsrc[radius], refs, blobs)])[0]
    widths = [w, w // 2, w // 2]
    got = ["%08x" % oracle.fnv1a(mv.plane_to_numpy(out[p], widths[p], gsup.dtype)) for p in range(3)]
    assert got == case["expect"]["degrain_fnv"]
    out = mv.Compensate(gsup, gan.ad).run([(gsf[radius], refs[0], blobs[0])])[0]
    got = ["%08x" % oracle.fnv1a(mv.plane_to_numpy(out[p], widths[p], gsup.dtype)) for p in range(3)]
    assert got == case["expect"]["compensate_fnv"]
    # Finest, Recalculate + divide, BlockFPS
    i = gsup.info
    fw_, fh_ = (w + 2 * i.hpad) * i.pel, (h + 2 * i.vpad) * i.pel
    fin = gsup.finest([gsf[radius]])[0]
    fwid = [fw_, fw_ // 2, fw_ // 2]
    assert ["%08x" % oracle.fnv1a(mv.plane_to_numpy(fin[p], fwid[p], gsup.dtype)) for p in range(3)] == case["expect"]["finest_fnv"]
    nf = len(frames)
    gabw = mv.Analyse(gsup, isb=1, delta=1, **c["akw"])
    gafw = mv.Analyse(gsup, isb=0, delta=1, **c["akw"])
    gbbw = gabw.run([(gsf[n], gsf[n + 1] if n + 1 < nf else None) for n in range(nf)])
    gbfw = gafw.run([(gsf[n], gsf[n - 1] if n >= 1 else None) for n in range(nf)])
    grc = mv.Recalculate(gsup, gabw.ad, blksize=8, overlap=4, thsad=100, divide=2)
    assert "%08x" % oracle.fnv1a(grc.run([(gsf[0], gsf[1], gbbw[0])])[0].cpu().numpy()) == case["expect"]["recalculate_divide_fnv"]
    for mode in (3, 7):
        gbf = mv.BlockFPS(gsup, gabw.ad, gafw.ad, nf, [p.stride(0) for p in gsrc[0]], 24, 1, num=60, den=1, mode=mode, ml=60.0)
        o = gbf.run([1], gsrc, gsf, gbbw, gbfw)[0]
        assert ["%08x" % oracle.fnv1a(mv.plane_to_numpy(o[p], widths[p], gsup.dtype)) for p in range(3)] == case["expect"]["blockfps_mode%d_fnv" % mode]


def test_full_size_properties_cfg2(mv):
    """BASELINE cfg2 at full size (1080p P8 Degrain1 blk 8 ov 4 pel 2) through size-independent properties:
    (1) a clip of identical frames has all-zero vectors and Degrain returns the input unchanged; (2) a pure integer
    translation is recovered exactly by the interior blocks; (3) the blob header / validity / sizes are self-consistent."""
    import torch
    w, h, bits = 1920, 1080, 8
    base = pl.moving_clip(w + 16, h + 16, bits, 1, seed=8, noise=0)[0]
    f0 = [base[0][8:8 + h, 8:8 + w], base[1][4:4 + h // 2, 4:4 + w // 2], base[2][4:4 + h // 2, 4:4 + w // 2]]
    f1 = [base[0][8:8 + h, 6:6 + w], base[1][4:4 + h // 2, 3:3 + w // 2], base[2][4:4 + h // 2, 3:3 + w // 2]]  # content moves +2 px
    sup = mv.Super(w, h, bits)
    src = [mv.frame_to_device([np.ascontiguousarray(p) for p in f]) for f in (f0, f0, f1)]
    sf = sup.build(src)
    an = mv.Analyse(sup, blksize=8, overlap=4, search=4, isb=1)
    b_same, b_shift = an.run([(sf[0], sf[1]), (sf[0], sf[2])])
    torch.cuda.synchronize()
    bs = b_same.cpu().numpy()
    assert bs[:8].view(np.int32).tolist() == [an.blob_size, 1] and an.blob_size == 2738792
    x, y, sad = pl.blob_vectors(bs, an.ad, 0)
    assert not x.any() and not y.any() and not sad.any()
    x, y, sad = pl.blob_vectors(b_shift.cpu().numpy(), an.ad, 0)
    inner = (slice(4, -4), slice(4, -4))
    assert (x[inner] == 4).mean() > 0.99 and (y[inner] == 0).mean() > 0.99  # +2 px == +4 half-pel
    out = mv.Degrain(1, sup, an.ad, [p.stride(0) for p in src[0]]).run([(src[0], [sf[1], sf[1]], [b_same, b_same])])[0]
    for p in range(3):
        assert torch.equal(out[p][:, :f0[p].shape[1]], src[0][p][:, :f0[p].shape[1]])


def _fullsize_props(mv, w, h, bits, blk, ov, tr, label):
    """Size-independent properties at a BASELINE size: (1) identical frames -> all-zero vectors, Degrain == input;
    (2) an integer translation is recovered by the interior blocks; (3) batch invariance: a chain gives the same blob
    alone and inside a multi-chain launch, forward / backward and with a missing reference; (4) blob header and size."""
    import torch
    scale = 1 << (bits - 8)
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:h + 16, 0:w + 16].astype(np.float32)
    tex = (40 * np.sin(xx * 0.21 + yy * 0.07) + 30 * np.sin(xx * 0.05 - yy * 0.13) + 25 * (((xx.astype(np.int32) // 8) + (yy.astype(np.int32) // 8)) & 1) + 120)
    tex = (tex + rng.integers(-2, 3, tex.shape)).clip(0, 255)
    dt = np.uint8 if bits == 8 else np.uint16
    big = [(tex * scale).astype(dt), (tex[::2, ::2] * 0.5 * scale + 64 * scale).astype(dt), (tex[::2, ::2] * 0.25 * scale + 96 * scale).astype(dt)]
    f0 = [big[0][8:8 + h, 8:8 + w], big[1][4:4 + h // 2, 4:4 + w // 2], big[2][4:4 + h // 2, 4:4 + w // 2]]
    f1 = [big[0][8:8 + h, 6:6 + w], big[1][4:4 + h // 2, 3:3 + w // 2], big[2][4:4 + h // 2, 3:3 + w // 2]]  # content moves +2 px
    sup = mv.Super(w, h, bits)
    src = [mv.frame_to_device([np.ascontiguousarray(p) for p in f]) for f in (f0, f0, f1)]
    sf = sup.build(src)
    an = mv.Analyse(sup, blksize=blk, overlap=ov, isb=1)
    jobs = [(sf[0], sf[1]), (sf[0], sf[2]), (sf[2], sf[0]), (sf[0], None)]
    batch = an.run(jobs)
    torch.cuda.synchronize()
    singles = [an.run([j])[0] for j in jobs]
    torch.cuda.synchronize()
    for b, s in zip(batch, singles):
        assert torch.equal(b, s), label + ": batch invariance"
    bs = batch[0].cpu().numpy()
    assert bs[:8].view(np.int32).tolist() == [an.blob_size, 1], label
    assert an.blob_size == mv.lib().mvx_vectors_size(C.byref(an.ad))
    x, y, sad = pl.blob_vectors(bs, an.ad, 0)
    assert not x.any() and not y.any() and not sad.any(), label + ": identical frames"
    x, y, sad = pl.blob_vectors(batch[1].cpu().numpy(), an.ad, 0)
    inner = (slice(4, -4), slice(4, -4))
    assert (x[inner] == 4).mean() > 0.99 and (y[inner] == 0).mean() > 0.99, label + ": +2 px == +4 half-pel"
    x, y, sad = pl.blob_vectors(batch[2].cpu().numpy(), an.ad, 0)
    assert (x[inner] == -4).mean() > 0.99 and (y[inner] == 0).mean() > 0.99, label + ": reverse direction"
    assert batch[3].cpu().numpy()[:8].view(np.int32).tolist() == [an.blob_size, 0], label + ": missing reference -> invalid blob"
    dg = mv.Degrain(tr, sup, an.ad, [p.stride(0) for p in src[0]])
    out = dg.run([(src[0], [sf[1]] * (2 * tr), [batch[0]] * (2 * tr))])[0]
    for p in range(3):
        # every block blends to the source value v; the overlap sum is sum_k((v * win_k) >> 6) with integer windows whose
        # taps add up to 2048 +- 1 (Overlap.cpp:40-125 rounds each tap), so the result is v +- a few v / 2048
        a = out[p].cpu().numpy()
        a = (a.view(np.uint16) if bits > 8 else a)[:, :f0[p].shape[1]].astype(np.int64)
        d = f0[p].astype(np.int64) - a
        assert (np.abs(d) <= (f0[p].astype(np.int64) >> 10) + 2).all(), label + ": Degrain of identical frames (%d..%d)" % (d.min(), d.max())
        assert (d == 0).mean() > 0.5, label + ": %.3f exact" % (d == 0).mean()


def _fullsize_parity(mv, oracle, w, h, bits, tr, akw, nout, replicas, want_k, label):
    """BASELINE configuration at FULL size against the oracle, byte for byte: `nout` consecutive output frames -- all 2*tr vector
    blobs and the three DegrainN planes of each -- with the searches launched inside a batch large enough to take the build the
    benchmark times (want_k chains per SIMD, a workgroup barrier every few blocks, shadow planes for 16-bit clips).  The batch is
    made of `replicas` copies of the distinct chains, each writing its own blob: every copy must equal the oracle's blob, so the
    test also covers what a chain's neighbours in the launch do to it.  MVAnalyse.c:189-239, MVDegrains.cpp:210-306."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    n = nout + 2 * tr
    frames = pl.moving_clip(w, h, bits, n, seed=21, noise=2)
    osup, gsup = oracle.Super(w, h, bits), mv.Super(w, h, bits)
    clips = [(d, isb) for d in range(1, tr + 1) for isb in (1, 0)]
    oan = {k: oracle.Analyse(osup, isb=k[1], delta=k[0], **akw) for k in clips}
    gan = mv.Analyse(gsup, **akw)  # (delta / isb only pick the reference frame: one parameter block, one launch, as bench.py does)
    gsrc = [mv.frame_to_device(f) for f in frames]
    gsf = gsup.build(gsrc)
    chains = [(f, f + d if isb else f - d) for f in range(tr, tr + nout) for d, isb in clips]
    jobs = [(gsf[a], gsf[b]) for a, b in chains] * replicas
    blobs = gan.run(jobs)
    torch.cuda.synchronize()
    info = (C.c_int * 5)()
    mv.lib().mvx_debug_last_launch(info)
    # (8-bit clips through the speculative kernel -- info[4] == 2 -- run without a barrier between a workgroup's chains)
    assert info[0] == want_k and (info[2] > 0 or (bits == 8 and info[4] == 2)) and info[3] == len(jobs), label + ": the batch did not take the %d-per-SIMD build with a barrier interval (%s)" % (want_k, list(info))
    with ThreadPoolExecutor(16) as ex:
        osf = list(ex.map(osup.frame, frames))
        oblobs = list(ex.map(lambda c: oan[(abs(c[1] - c[0]), 1 if c[1] > c[0] else 0)].frame(osf[c[0]], osf[c[1]]), chains))
    nc = len(chains)
    want = [torch.from_numpy(b).to(blobs[0].device) for b in oblobs]
    for i, b in enumerate(blobs):
        assert torch.equal(b, want[i % nc]), label + ": vectors of chain %s differ from the oracle (copy %d of %d)" % (chains[i % nc], i // nc, replicas)
    # r5: the same chains as a SMALL launch -- the library runs it as teams (the waves of a workgroup walk one chain) -- must give the same blobs
    small = gan.run([(gsf[a], gsf[b]) for a, b in chains])
    torch.cuda.synchronize()
    mv.lib().mvx_debug_last_launch(info)
    assert info[4] == 3 and info[1] == 4, label + ": the small launch did not take the team form (%s)" % list(info)
    for i, b in enumerate(small):
        assert torch.equal(b, want[i]), label + ": vectors of chain %s differ from the oracle in the team form" % (chains[i],)
    gdg = mv.Degrain(tr, gsup, gan.ad, [p.stride(0) for p in gsrc[0]])
    odg = oracle.Degrain(tr, osup, oan[clips[0]].ad)
    for k, f in enumerate(range(tr, tr + nout)):
        refs = [f + d if isb else f - d for d, isb in clips]
        fb = blobs[k * len(clips):(k + 1) * len(clips)]
        got = gdg.run([(gsrc[f], [gsf[r] for r in refs], fb)])[0]
        exp = odg.frame(frames[f], [osf[r] for r in refs], oblobs[k * len(clips):(k + 1) * len(clips)])
        for p in range(3):
            assert np.array_equal(mv.plane_to_numpy(got[p], exp[p].shape[1], exp[p].dtype), exp[p]), label + ": Degrain%d output frame %d plane %d" % (tr, f, p)


def test_full_size_parity_cfg3(mv, oracle):
    """BASELINE cfg3 (4K YUV420P16 Degrain3 blk 16 ov 8 pel 2), full size, byte for byte, inside a 2 040-chain launch = the two
    chains per SIMD build of the speculative kernel that bench.py times (r4)"""
    _fullsize_parity(mv, oracle, 3840, 2160, 16, 3, dict(blksize=16, overlap=8), nout=2, replicas=170, want_k=2, label="cfg3")


def test_full_size_parity_cfg2(mv, oracle):
    """BASELINE cfg2 (1080p YUV420P8 Degrain1 blk 8 ov 4 pel 2 search 4), full size, inside a 2 046-chain launch (the speculative kernel, two per SIMD)"""
    _fullsize_parity(mv, oracle, 1920, 1080, 8, 1, dict(blksize=8, overlap=4, search=4), nout=3, replicas=341, want_k=2, label="cfg2")  # (2046 chains: the speculative kernel's two per SIMD in one round)


def test_full_size_parity_cfg5(mv, oracle):
    """BASELINE cfg5 (8K YUV420P16 Degrain6 blk 32 ov 16 pel 2), full size, inside a 1 032-chain launch (two per SIMD)"""
    _fullsize_parity(mv, oracle, 7680, 4320, 16, 6, dict(blksize=32, overlap=16), nout=1, replicas=86, want_k=2, label="cfg5")


@pytest.mark.gpu
def test_full_size_properties_cfg3(mv):
    """BASELINE cfg3: 4K YUV420P16, blk 16, overlap 8, pel 2, Degrain3."""
    _fullsize_props(mv, 3840, 2160, 16, 16, 8, 3, "cfg3")


@pytest.mark.gpu
def test_full_size_properties_cfg5(mv):
    """BASELINE cfg5: 8K YUV420P16, blk 32, overlap 16, pel 2, Degrain6."""
    _fullsize_props(mv, 7680, 4320, 16, 32, 16, 6, "cfg5")


@pytest.mark.gpu
def test_full_size_properties_cfg4(mv):
    """BASELINE cfg4: 1080p YUV420P8, blk 8, pel 2, mv.Compensate + mv.BlockFPS 24 -> 60 at full size, through properties that do
    not depend on the size: identical frames interpolate to themselves; a pure +2 px translation is compensated exactly and the
    frame half way between (24 -> 48) is the content moved by 1 px; time positions 0 copy the input frame."""
    import torch
    w, h, bits, blk = 1920, 1080, 8, 8
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:h + 16, 0:w + 16].astype(np.float32)
    tex = (40 * np.sin(xx * 0.21 + yy * 0.07) + 30 * np.sin(xx * 0.05 - yy * 0.13) + 25 * (((xx.astype(np.int32) // 8) + (yy.astype(np.int32) // 8)) & 1) + 120)
    tex = (tex + rng.integers(-2, 3, tex.shape)).clip(0, 255)
    big = [tex.astype(np.uint8), (tex[::2, ::2] * 0.5 + 64).astype(np.uint8), (tex[::2, ::2] * 0.25 + 96).astype(np.uint8)]
    cut = lambda dx: [np.ascontiguousarray(big[0][8:8 + h, 8 - dx:8 - dx + w]), np.ascontiguousarray(big[1][4:4 + h // 2, 4 - dx // 2:4 - dx // 2 + w // 2]),
                      np.ascontiguousarray(big[2][4:4 + h // 2, 4 - dx // 2:4 - dx // 2 + w // 2])]
    f = [cut(0), cut(0), cut(2), cut(4)]   # frames 0 == 1, then the content moves +2 px per frame
    nf = len(f)
    sup = mv.Super(w, h, bits)
    src = [mv.frame_to_device(x) for x in f]
    sf = sup.build(src)
    abw = mv.Analyse(sup, num_frames=nf, blksize=blk, isb=1)
    afw = mv.Analyse(sup, num_frames=nf, blksize=blk, isb=0)
    bbw = abw.run([(sf[n], sf[n + 1] if n + 1 < nf else None) for n in range(nf)])
    bfw = afw.run([(sf[n], sf[n - 1] if n >= 1 else None) for n in range(nf)])
    torch.cuda.synchronize()
    plane = lambda t, p: t[p].cpu().numpy()[:, :f[0][p].shape[1]]
    # Compensate: frame 2 fetched with frame 1's backward vectors reproduces frame 1 in the interior (integer motion)
    comp = mv.Compensate(sup, abw.ad).run([(sf[1], sf[2], bbw[1]), (sf[0], sf[1], bbw[0])])
    torch.cuda.synchronize()
    for p in range(3):
        m = 24 if p == 0 else 12
        assert np.array_equal(plane(comp[0], p)[m:-m, m:-m], f[1][p][m:-m, m:-m]), "cfg4: compensated translation, plane %d" % p
        assert np.array_equal(plane(comp[1], p), f[0][p]), "cfg4: compensation of identical frames, plane %d" % p
    # BlockFPS 24 -> 48: output 2n copies input n; output 1 lies between the identical frames 0 and 1; output 5 half way between 2 and 3
    fps = mv.BlockFPS(sup, abw.ad, afw.ad, nf, [t.stride(0) for t in src[0]], 24, 1, num=48, den=1)
    assert fps.map(5) == (2, 3, 128) and fps.map(4)[2] == 0
    out = fps.run([0, 1, 4, 5], src, sf, bbw, bfw)
    torch.cuda.synchronize()
    mid = cut(3)
    for p in range(3):
        assert np.array_equal(plane(out[0], p), f[0][p]) and np.array_equal(plane(out[2], p), f[2][p]), "cfg4: time position 0 copies the frame"
        assert np.array_equal(plane(out[1], p), f[0][p]), "cfg4: identical frames interpolate to themselves, plane %d" % p
        if p == 0:  # (+3 px is not a whole chroma sample: luma only)
            assert np.array_equal(plane(out[3], 0)[32:-32, 32:-32], mid[0][32:-32, 32:-32]), "cfg4: half-way frame of a +2 px translation"
    # BlockFPS 24 -> 60 (the BASELINE rate): runs at full size, time positions as the reference's arithmetic gives them
    fps60 = mv.BlockFPS(sup, abw.ad, afw.ad, nf, [t.stride(0) for t in src[0]], 24, 1, num=60, den=1)
    assert fps60.num_frames == 1 + (nf - 1) * 60 // 24 and [fps60.map(k)[:2] for k in (0, 1, 2, 3, 5)] == [(0, 1), (0, 1), (0, 1), (1, 2), (2, 3)]  # MVBlockFPS.c:955
    out60 = fps60.run(list(range(5)), src, sf, bbw, bfw)
    torch.cuda.synchronize()
    for k in (1, 2):
        assert np.array_equal(plane(out60[k], 0), f[0][0]), "cfg4: 24 -> 60 between identical frames"


BLOCKFPS_CASES = [
    # w, h, bits, analyse kwargs, blockfps kwargs (24 fps input)
    (128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1)),                       # BASELINE cfg4: 24 -> 60
    (128, 96, 8, dict(blksize=8, overlap=0), dict(num=60, den=1)),
    (192, 112, 16, dict(blksize=16, overlap=8), dict(num=48, den=1, mode=0)),
    (200, 120, 8, dict(blksize=8, overlap=4), dict(num=60, den=1, mode=1)),              # uncovered strips
    (200, 120, 8, dict(blksize=8, overlap=0), dict(num=60, den=1, mode=2)),
    (128, 96, 16, dict(blksize=8, overlap=4), dict(num=60, den=1, mode=4, ml=40.0)),
    (128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1, mode=5, ml=20.0)),
    (128, 96, 8, dict(blksize=16, overlap=8), dict(num=60, den=1, mode=6, ml=50.0)),
    (128, 96, 16, dict(blksize=8, overlap=2), dict(num=60, den=1, mode=7, ml=50.0)),
    (128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1, mode=8, ml=30.0)),
    (128, 96, 8, dict(blksize=8, overlap=4), dict(num=0, den=0)),                        # default: double rate
    (128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1, thscd1=20, thscd2=10)), # scene change -> blend fallback
    (128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1, thscd1=20, thscd2=10, blend=0)),
    (128, 96, 8, dict(blksize=8, overlap=4, delta=2), dict(num=36, den=1)),               # delta 2
]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bits,akw,bkw", BLOCKFPS_CASES)
def test_blockfps_parity(oracle, mv, w, h, bits, akw, bkw):
    import torch
    nf = 6
    akw = dict(akw)
    delta = akw.pop("delta", 1)
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, {}, akw, nframes=nf, seed=41)
    oabw = oracle.Analyse(osup, num_frames=nf, isb=1, delta=delta, **akw)
    oafw = oracle.Analyse(osup, num_frames=nf, isb=0, delta=delta, **akw)
    obbw = [oabw.frame(osf[n], osf[n + delta] if n + delta < nf else None) for n in range(nf)]
    obfw = [oafw.frame(osf[n], osf[n - delta] if n - delta >= 0 else None) for n in range(nf)]
    gabw = mv.Analyse(gsup, num_frames=nf, isb=1, delta=delta, **akw)
    gafw = mv.Analyse(gsup, num_frames=nf, isb=0, delta=delta, **akw)
    gbbw = gabw.run([(gsf[n], gsf[n + delta] if n + delta < nf else None) for n in range(nf)])
    gbfw = gafw.run([(gsf[n], gsf[n - delta] if n - delta >= 0 else None) for n in range(nf)])
    ob = oracle.BlockFPS(osup, oabw.ad, oafw.ad, nf, 24, 1, **bkw)
    gb = mv.BlockFPS(gsup, gabw.ad, gafw.ad, nf, [p.stride(0) for p in gsrc[0]], 24, 1, **bkw)
    assert gb.num_frames == ob.num_frames and (gb.fps_num, gb.fps_den) == (ob.d.outFpsNum, ob.d.outFpsDen)
    ns = list(range(gb.num_frames))
    for n in ns:
        assert gb.map(n) == ob.map(n)
    out = gb.run(ns, gsrc, gsf, gbbw, gbfw)
    torch.cuda.synchronize()
    for n in ns:
        want = ob.frame(n, frames, osf, obbw, obfw)
        for p in range(3):
            got = out[n][p].cpu().numpy()
            got = (got.view(np.uint16) if bits > 8 else got)[:, :want[p].shape[1]]
            assert np.array_equal(got, want[p]), (n, p, gb.map(n), int(np.count_nonzero(got != want[p])))


@pytest.mark.gpu
@pytest.mark.parametrize("bits,akw", [(8, dict(blksize=16, overlap=8, divide=1)), (8, dict(blksize=16, overlap=8, divide=2)), (16, dict(blksize=8, overlap=4, divide=2)),
                                      (8, dict(blksize=16, overlap=0, divide=2))])
def test_analyse_divide_parity_and_degrain_on_divided_vectors(oracle, mv, bits, akw):
    """divide = 1 / 2 (GroupOfPlanes.c:206-302): blob with the extra array of half-size blocks, the divided analysis data, and a
    Degrain that reads such a vector clip (readers find level 0 by walking the plane size headers)."""
    import torch
    w, h, nf = 192, 128, 3
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, {}, {}, nframes=nf, seed=51)
    oan = {isb: oracle.Analyse(osup, num_frames=nf, isb=isb, **akw) for isb in (1, 0)}
    gan = {isb: mv.Analyse(gsup, num_frames=nf, isb=isb, **akw) for isb in (1, 0)}
    for k, _ in oracle.AnalysisData._fields_:
        if k not in ("nMagicKey", "nVersion", "nCPUFlags"):
            assert getattr(gan[1].ad, k) == getattr(oan[1].ad, k), k
    assert gan[1].blob_size == oan[1].blob_size
    want = {1: oan[1].frame(osf[1], osf[2]), 0: oan[0].frame(osf[1], osf[0])}
    got = {1: gan[1].run([(gsf[1], gsf[2])])[0], 0: gan[0].run([(gsf[1], gsf[0])])[0]}
    inval_w, inval_g = oan[1].frame(osf[2], None), gan[1].run([(gsf[2], None)])[0]
    torch.cuda.synchronize()
    for isb in (1, 0):
        assert np.array_equal(got[isb].cpu().numpy(), want[isb]), isb
    assert np.array_equal(inval_g.cpu().numpy(), inval_w)
    odg = oracle.Degrain(1, osup, oan[1].ad)
    gdg = mv.Degrain(1, gsup, gan[1].ad, [p.stride(0) for p in gsrc[0]])
    wout = odg.frame(frames[1], [osf[2], osf[0]], [want[1], want[0]])
    gout = gdg.run([(gsrc[1], [gsf[2], gsf[0]], [got[1], got[0]])])[0]
    for p in range(3):
        g = gout[p].cpu().numpy()
        g = (g.view(np.uint16) if bits > 8 else g)[:, :wout[p].shape[1]]
        assert np.array_equal(g, wout[p]), p


RECALC_CASES = [
    # bits, old analyse kwargs, recalculate kwargs
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100)),
    (16, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=50, smooth=0)),
    (8, dict(blksize=16, overlap=0), dict(blksize=8, overlap=2, thsad=0, search=3, searchparam=2)),
    (8, dict(blksize=8, overlap=4), dict(blksize=16, overlap=8, thsad=80, search=5, searchparam=4)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=60, search=0, searchparam=4)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=60, search=1, searchparam=3)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=60, search=2, searchparam=4)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=60, search=6, searchparam=3)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=60, search=7, searchparam=3)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100, chroma=0, truemotion=0)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100, divide=2)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100, dct=5)),
    (16, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100, dct=6)),
    (8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100, dct=7)),
    (8, dict(blksize=16, overlap=8, divide=2), dict(blksize=8, overlap=4, thsad=100)),   # refining a divided clip
]


@pytest.mark.gpu
@pytest.mark.parametrize("bits,akw,rkw", RECALC_CASES)
def test_recalculate_parity(oracle, mv, bits, akw, rkw):
    import torch
    w, h, nf = 192, 128, 3
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, dict(pel=2), {}, nframes=nf, seed=53)
    oan = oracle.Analyse(osup, num_frames=nf, isb=1, **akw)
    gan = mv.Analyse(gsup, num_frames=nf, isb=1, **akw)
    oold = [oan.frame(osf[n], osf[n + 1] if n + 1 < nf else None) for n in range(nf)]
    gold = gan.run([(gsf[n], gsf[n + 1] if n + 1 < nf else None) for n in range(nf)])
    orc = oracle.Recalculate(osup, oan.ad, **rkw)
    grc = mv.Recalculate(gsup, gan.ad, **rkw)
    assert grc.blob_size == orc.blob_size
    for k, _ in oracle.AnalysisData._fields_:
        if k not in ("nMagicKey", "nVersion", "nCPUFlags"):
            assert getattr(grc.ad, k) == getattr(orc.ad, k), k
    got = grc.run([(gsf[n], gsf[n + 1] if n + 1 < nf else None, gold[n]) for n in range(nf)])
    torch.cuda.synchronize()
    for n in range(nf):
        want = orc.frame(osf[n], osf[n + 1] if n + 1 < nf else None, oold[n])
        g = got[n].cpu().numpy()
        assert np.array_equal(g, want), (n, int(np.count_nonzero(g != want)))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bits,skw", [(128, 96, 8, {}), (200, 120, 16, dict(pel=4)), (128, 96, 8, dict(pel=1)), (128, 96, 16, dict(chroma=0)), (132, 100, 8, dict(hpad=8, vpad=4))])
def test_finest_parity(oracle, mv, w, h, bits, skw):
    import torch
    frames = pl.moving_clip(w, h, bits, 2, seed=61, noise=3)
    osup = oracle.Super(w, h, bits, **skw)
    gsup = mv.Super(w, h, bits, **skw)
    gsf = gsup.build([mv.frame_to_device(f) for f in frames])
    out = gsup.finest(gsf)
    torch.cuda.synchronize()
    for n in range(2):
        want = osup.finest(osup.frame(frames[n]))
        for p in range(3):
            if skw.get("chroma", 1) == 0 and p:
                continue  # not written (MVFinest.c:87)
            g = out[n][p].cpu().numpy()
            g = (g.view(np.uint16) if bits > 8 else g)[:, :want[p].shape[1]]
            assert np.array_equal(g, want[p]), (n, p)


@pytest.mark.gpu
def test_scdetect_matches_oracle(oracle, mv):
    import torch
    w, h, bits, nf = 128, 96, 8, 4
    frames, osup, gsup, osf, gsrc, gsf = _pipeline(oracle, mv, w, h, bits, 1, {}, {}, nframes=nf, seed=63)
    frames2 = pl.moving_clip(w, h, bits, 1, seed=999, noise=30)  # an unrelated frame = a scene change
    gcut = gsup.build([mv.frame_to_device(frames2[0])])[0]
    ocut = osup.frame(frames2[0])
    akw = dict(blksize=8, overlap=4)
    oan = oracle.Analyse(osup, num_frames=nf, isb=1, **akw)
    gan = mv.Analyse(gsup, num_frames=nf, isb=1, **akw)
    gjobs = [(gsf[0], gsf[1]), (gsf[1], gcut), (gsf[2], None)]
    ojobs = [(osf[0], osf[1]), (osf[1], ocut), (osf[2], None)]
    gb = gan.run(gjobs)
    for th in (dict(), dict(thscd1=100, thscd2=40), dict(thscd1=10, thscd2=5)):
        got = mv.scdetect(gan.ad, gb, **th)
        t1, t2 = C.c_int64(th.get("thscd1", 400)), C.c_int(th.get("thscd2", 130))
        oracle.lib().mvo_scale_thscd(C.byref(t1), C.byref(t2), C.byref(oan.d.ad))
        want = []
        for s, r in ojobs:
            b = oan.frame(s, r)
            want.append(int(not oracle.lib().mvo_blob_is_usable(C.byref(oan.d.ad), C.c_void_p(b.ctypes.data), t1.value, t2.value)))
        assert got == want, (th, got, want)
    assert mv.scdetect(gan.ad, gb)[2] == 1  # invalid vectors count as a scene change
    with pytest.raises(mv.MvtoolsError):
        mv.scdetect(gan.ad, gb, thscd1=8 * 8 * 255 + 1)


def test_staged_copies_round_trip(mv):
    """mvx_upload_2d / mvx_download_2d (pinned staging inside the library) from several threads at once: rows of odd lengths, host and
    device pitches that differ, more concurrent callers than staging buffers"""
    import threading
    L = mv.lib()
    L.mvx_upload_2d.argtypes = L.mvx_download_2d.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_size_t, C.c_size_t, C.c_void_p]
    L.mvx_dev_alloc_uninit.restype = C.c_void_p
    L.mvx_dev_alloc_uninit.argtypes = [C.c_size_t]
    L.mvx_dev_free.argtypes = [C.c_void_p]
    L.mvx_stream_create_priority.restype = C.c_void_p
    L.mvx_stream_create_priority.argtypes = [C.c_int]
    L.mvx_stream_destroy.argtypes = [C.c_void_p]
    stream = L.mvx_stream_create_priority(1)
    assert stream
    errors = []

    def work(i):
        rng = np.random.default_rng(100 + i)
        rows, rb = 37 + 11 * i, 1001 + 333 * i
        hp, dp, hp2 = rb + 7, (rb + 255) // 256 * 256, rb + 64
        src = rng.integers(0, 256, (rows, hp), dtype=np.uint8)
        dst = np.zeros((rows, hp2), dtype=np.uint8)
        dev = L.mvx_dev_alloc_uninit(rows * dp)
        try:
            if L.mvx_upload_2d(dev, dp, src.ctypes.data, hp, rb, rows, stream) or L.mvx_download_2d(dst.ctypes.data, hp2, dev, dp, rb, rows, stream):
                errors.append("call %d failed" % i)
            elif not np.array_equal(dst[:, :rb], src[:, :rb]) or dst[:, rb:].any():
                errors.append("data %d differs" % i)
        finally:
            L.mvx_dev_free(dev)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(24)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    L.mvx_stream_destroy(stream)
    assert not errors, errors


@pytest.mark.gpu
def test_staged_copies_row_longer_than_a_staging_buffer(mv):
    """a single row of more than 16 MiB (the shell moves whole vector blobs as one row: 4K with 8x8 blocks and divide, 8K with 8x8 blocks) goes
    through the staging buffers segment by segment (such rows were rejected)"""
    L = mv.lib()
    L.mvx_upload_2d.argtypes = L.mvx_download_2d.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_size_t, C.c_size_t, C.c_void_p]
    L.mvx_dev_alloc_uninit.restype = C.c_void_p
    L.mvx_dev_alloc_uninit.argtypes = [C.c_size_t]
    L.mvx_dev_free.argtypes = [C.c_void_p]
    L.mvx_stream_create_priority.restype = C.c_void_p
    L.mvx_stream_create_priority.argtypes = [C.c_int]
    L.mvx_stream_destroy.argtypes = [C.c_void_p]
    stream = L.mvx_stream_create_priority(1)
    rb = (44 << 20) + 12345  # 44 MiB and a bit: two full segments and a tail
    rows = 2
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (rows, rb + 3), dtype=np.uint8)
    dst = np.zeros((rows, rb + 5), dtype=np.uint8)
    dp = (rb + 255) // 256 * 256
    dev = L.mvx_dev_alloc_uninit(rows * dp)
    try:
        assert L.mvx_upload_2d(dev, dp, src.ctypes.data, src.strides[0], rb, rows, stream) == 0, mv.last_error() if hasattr(mv, "last_error") else "upload failed"
        assert L.mvx_download_2d(dst.ctypes.data, dst.strides[0], dev, dp, rb, rows, stream) == 0
        assert np.array_equal(dst[:, :rb], src[:, :rb]) and not dst[:, rb:].any()
    finally:
        L.mvx_dev_free(dev)
        L.mvx_stream_destroy(stream)
