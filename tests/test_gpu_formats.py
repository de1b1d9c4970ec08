"""GPU parity on the chroma formats other than 4:2:0 -- 4:4:4, 4:2:2 and Gray, 8 and 16 bit -- for every filter of the path.

The reference accepts them in every filter (MVAnalyse.c:463-517, MVDegrains.cpp:161-162,342,693,776, MVCompensate.c:128-129,543,
MVBlockFPS.c:326-327, MVRecalculate.c:370-400): the chroma block is blk / xRatioUV by blk / yRatioUV, a Gray clip has one plane and
searches luma only.  In the library these formats take the general search kernel (mvx_fast_eligible wants xr == yr == 2) and the
run-time chroma ratios of degrain_kernel / compensate_kernel / blockfps_kernel; round 5's suite exercised them through mv.Super only
("format gap").  Bit-exact, through the C ABI, against the oracle on the same seeded clips."""
import numpy as np
import pytest

import pipeline as pl

pytestmark = pytest.mark.gpu

FORMATS = {
    "444": dict(subsampling=(0, 0)),
    "422": dict(subsampling=(1, 0)),
    "gray": dict(gray=True),
}


def _clip(w, h, bits, n, fmt, seed, noise=3):
    f = FORMATS[fmt]
    frames = pl.moving_clip(w, h, bits, n, seed=seed, noise=noise, sub=f.get("subsampling", (1, 1)))
    return [[fr[0]] for fr in frames] if f.get("gray") else frames


def _setup(oracle, mv, w, h, bits, n, fmt, skw, seed):
    frames = _clip(w, h, bits, n, fmt, seed)
    kw = dict(FORMATS[fmt], **skw)
    osup, gsup = oracle.Super(w, h, bits, **kw), mv.Super(w, h, bits, **kw)
    osf = [osup.frame(f) for f in frames]
    gsrc = [mv.frame_to_device(f) for f in frames]
    gsf = gsup.build(gsrc)
    return frames, osup, gsup, osf, gsrc, gsf


def _same_ad(oracle, g, o):
    for k, _ in oracle.AnalysisData._fields_:
        if k not in ("nMagicKey", "nVersion", "nCPUFlags"):  # never initialised / host dependent in the reference
            assert getattr(g, k) == getattr(o, k), k


def _planes_equal(mv, got, want, what):
    assert len(got) >= len(want)
    for p in range(len(want)):
        g = mv.plane_to_numpy(got[p], want[p].shape[1], want[p].dtype)
        if not np.array_equal(g, want[p]):
            ys, xs = np.nonzero(g != want[p])
            pytest.fail("%s plane %d: %d samples differ, first (y=%d,x=%d) gpu %d oracle %d" % (what, p, len(ys), ys[0], xs[0], g[ys[0], xs[0]], want[p][ys[0], xs[0]]))


ANALYSE_FMT_CASES = [
    # fmt, w, h, bits, super kwargs, analyse kwargs
    ("444", 192, 112, 8, {}, dict(blksize=8, overlap=4)),
    ("444", 192, 112, 16, {}, dict(blksize=16, overlap=8)),
    ("444", 200, 120, 8, dict(pel=4), dict(blksize=8, overlap=2, search=3, searchparam=2)),
    ("444", 192, 112, 8, {}, dict(blksize=8, overlap=4, dct=5)),                 # SATD on full-size chroma blocks
    ("444", 256, 144, 16, {}, dict(blksize=32, overlap=16)),
    ("422", 192, 112, 8, {}, dict(blksize=8, overlap=4)),
    ("422", 192, 112, 16, {}, dict(blksize=16, overlap=8)),
    ("422", 200, 120, 8, dict(pel=1), dict(blksize=16, blksizev=8, overlap=8, overlapv=4)),
    ("422", 192, 112, 16, {}, dict(blksize=8, overlap=4, dct=7)),                # chroma blocks 4 x 8
    ("422", 192, 112, 8, {}, dict(blksize=16, overlap=0, trymany=1)),
    ("gray", 192, 112, 8, {}, dict(blksize=8, overlap=4)),
    ("gray", 192, 112, 16, {}, dict(blksize=16, overlap=8)),
    ("gray", 200, 120, 8, dict(pel=4), dict(blksize=8, overlap=0, search=5, searchparam=4)),
    ("gray", 192, 112, 16, {}, dict(blksize=8, overlap=4, dct=6)),
    ("gray", 256, 144, 16, {}, dict(blksize=32, overlap=16, badsad=300, badrange=8)),
]


@pytest.mark.parametrize("fmt,w,h,bits,skw,akw", ANALYSE_FMT_CASES)
def test_analyse_formats(oracle, mv, fmt, w, h, bits, skw, akw):
    import torch
    frames, osup, gsup, osf, gsrc, gsf = _setup(oracle, mv, w, h, bits, 2, fmt, skw, seed=71)
    for isb in (1, 0):
        oan, gan = oracle.Analyse(osup, isb=isb, **akw), mv.Analyse(gsup, isb=isb, **akw)
        assert gan.blob_size == oan.blob_size
        _same_ad(oracle, gan.ad, oan.ad)
        got = gan.run([(gsf[0], gsf[1]), (gsf[1], gsf[0]), (gsf[0], None)])
        torch.cuda.synchronize()
        want = [oan.frame(osf[0], osf[1]), oan.frame(osf[1], osf[0]), oan.frame(osf[0], None)]
        for i in range(3):
            g = got[i].cpu().numpy()
            if not np.array_equal(g, want[i]):
                gx, gy, gs = pl.blob_vectors(g, oan.ad, 0)
                wx, wy, ws = pl.blob_vectors(want[i], oan.ad, 0)
                d = (gx != wx) | (gy != wy) | (gs != ws)
                pytest.fail("%s isb %d job %d: blob differs; level 0: %d / %d blocks" % (fmt, isb, i, int(d.sum()), d.size))


DEGRAIN_FMT_CASES = [
    # fmt, w, h, bits, radius, analyse kwargs, degrain kwargs
    ("444", 128, 96, 8, 1, dict(blksize=8, overlap=4), {}),
    ("444", 192, 112, 16, 3, dict(blksize=16, overlap=8), {}),
    ("444", 200, 120, 8, 2, dict(blksize=8, overlap=0), dict(limit=3, limitc=5)),
    ("422", 128, 96, 8, 1, dict(blksize=8, overlap=4), {}),
    ("422", 192, 112, 16, 2, dict(blksize=16, overlap=8), dict(plane=3)),
    ("422", 200, 120, 16, 1, dict(blksize=16, overlap=0), {}),
    ("422", 196, 116, 8, 1, dict(blksize=16, blksizev=8, overlap=8, overlapv=4), dict(thsadc=150)),
    ("gray", 128, 96, 8, 1, dict(blksize=8, overlap=4), {}),
    ("gray", 192, 112, 16, 3, dict(blksize=16, overlap=8), {}),
    ("gray", 200, 120, 8, 1, dict(blksize=8, overlap=0), dict(limit=2)),
]


@pytest.mark.parametrize("fmt,w,h,bits,radius,akw,dkw", DEGRAIN_FMT_CASES)
def test_degrain_formats(oracle, mv, fmt, w, h, bits, radius, akw, dkw):
    import torch
    frames, osup, gsup, osf, gsrc, gsf = _setup(oracle, mv, w, h, bits, 2 * radius + 1, fmt, {}, seed=73)
    n = len(frames)
    for target in (radius, 0):
        oblobs, gblobs, refs_o, refs_g = [], [], [], []
        for d in range(1, radius + 1):
            for isb in (1, 0):
                oan, gan = oracle.Analyse(osup, isb=isb, delta=d, **akw), mv.Analyse(gsup, isb=isb, delta=d, **akw)
                nref = target + (d if isb else -d)
                ok = 0 <= nref < n
                oblobs.append(oan.frame(osf[target], osf[nref] if ok else None))
                gblobs.append(gan.run([(gsf[target], gsf[nref] if ok else None)])[0])
                refs_o.append(osf[nref] if ok else None)
                refs_g.append(gsf[nref] if ok else None)
        for a, b in zip(gblobs, oblobs):
            assert np.array_equal(a.cpu().numpy(), b), "vectors differ"
        odg = oracle.Degrain(radius, osup, oan.ad, **dkw)
        gdg = mv.Degrain(radius, gsup, gan.ad, [p.stride(0) for p in gsrc[0]], **dkw)
        want = odg.frame(frames[target], refs_o, oblobs)
        got = gdg.run([(gsrc[target], refs_g, gblobs)])[0]
        torch.cuda.synchronize()
        _planes_equal(mv, got, want, "%s degrain%d target %d" % (fmt, radius, target))
        if target == radius:
            assert any(not np.array_equal(want[p], frames[target][p]) for p in range(len(want))), "the case must denoise something"


COMP_FMT_CASES = [
    ("444", 128, 96, 8, dict(blksize=8, overlap=4), {}),
    ("444", 192, 112, 16, dict(blksize=16, overlap=8), dict(thsad=60)),
    ("444", 200, 120, 8, dict(blksize=8, overlap=0), {}),
    ("422", 128, 96, 8, dict(blksize=8, overlap=4), dict(time=40.0)),
    ("422", 192, 112, 16, dict(blksize=16, overlap=8), {}),
    ("422", 200, 120, 16, dict(blksize=16, overlap=0), dict(scbehavior=0)),
    ("gray", 128, 96, 8, dict(blksize=8, overlap=4), {}),
    ("gray", 192, 112, 16, dict(blksize=16, overlap=8), dict(thscd1=20, thscd2=10)),
    ("gray", 200, 120, 16, dict(blksize=8, overlap=0), {}),
]


@pytest.mark.parametrize("fmt,w,h,bits,akw,ckw", COMP_FMT_CASES)
def test_compensate_formats(oracle, mv, fmt, w, h, bits, akw, ckw):
    import torch
    frames, osup, gsup, osf, gsrc, gsf = _setup(oracle, mv, w, h, bits, 2, fmt, {}, seed=75)
    oan, gan = oracle.Analyse(osup, isb=1, **akw), mv.Analyse(gsup, isb=1, **akw)
    for (src, ref) in ((0, 1), (1, None)):
        oref, gref = (osf[ref], gsf[ref]) if ref is not None else (None, None)
        ob = oan.frame(osf[src], oref)
        gb = gan.run([(gsf[src], gref)])[0]
        assert np.array_equal(gb.cpu().numpy(), ob)
        oc, gc = oracle.Compensate(osup, oan.ad, **ckw), mv.Compensate(gsup, gan.ad, **ckw)
        want = oc.frame(osf[src], oref, ob)
        got = gc.run([(gsf[src], gref, gb)])[0]
        torch.cuda.synchronize()
        _planes_equal(mv, got, want, "%s compensate src %d" % (fmt, src))


BLOCKFPS_FMT_CASES = [
    ("444", 128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1)),
    ("444", 192, 112, 16, dict(blksize=16, overlap=8), dict(num=48, den=1, mode=0)),
    ("422", 128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1, mode=5, ml=20.0)),
    ("422", 200, 120, 16, dict(blksize=16, overlap=0), dict(num=60, den=1, mode=2)),
    ("gray", 128, 96, 8, dict(blksize=8, overlap=4), dict(num=60, den=1)),
    ("gray", 192, 112, 16, dict(blksize=16, overlap=8), dict(num=60, den=1, mode=7, ml=50.0)),
]


@pytest.mark.parametrize("fmt,w,h,bits,akw,bkw", BLOCKFPS_FMT_CASES)
def test_blockfps_formats(oracle, mv, fmt, w, h, bits, akw, bkw):
    import torch
    nf = 5
    frames, osup, gsup, osf, gsrc, gsf = _setup(oracle, mv, w, h, bits, nf, fmt, {}, seed=77)
    oabw, oafw = oracle.Analyse(osup, num_frames=nf, isb=1, **akw), oracle.Analyse(osup, num_frames=nf, isb=0, **akw)
    gabw, gafw = mv.Analyse(gsup, num_frames=nf, isb=1, **akw), mv.Analyse(gsup, num_frames=nf, isb=0, **akw)
    obbw = [oabw.frame(osf[n], osf[n + 1] if n + 1 < nf else None) for n in range(nf)]
    obfw = [oafw.frame(osf[n], osf[n - 1] if n >= 1 else None) for n in range(nf)]
    gbbw = gabw.run([(gsf[n], gsf[n + 1] if n + 1 < nf else None) for n in range(nf)])
    gbfw = gafw.run([(gsf[n], gsf[n - 1] if n >= 1 else None) for n in range(nf)])
    ob = oracle.BlockFPS(osup, oabw.ad, oafw.ad, nf, 24, 1, **bkw)
    gb = mv.BlockFPS(gsup, gabw.ad, gafw.ad, nf, [p.stride(0) for p in gsrc[0]], 24, 1, **bkw)
    assert gb.num_frames == ob.num_frames
    ns = list(range(gb.num_frames))
    out = gb.run(ns, gsrc, gsf, gbbw, gbfw)
    torch.cuda.synchronize()
    for n in ns:
        assert gb.map(n) == ob.map(n)
        _planes_equal(mv, out[n], ob.frame(n, frames, osf, obbw, obfw), "%s blockfps frame %d %s" % (fmt, n, gb.map(n)))


RECALC_FMT_CASES = [
    ("444", 8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100)),
    ("444", 16, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100, dct=5)),
    ("422", 8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=60, search=3, searchparam=2)),
    ("422", 16, dict(blksize=16, overlap=0), dict(blksize=8, overlap=2, thsad=0)),
    ("gray", 8, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=100)),
    ("gray", 16, dict(blksize=16, overlap=8), dict(blksize=8, overlap=4, thsad=50, smooth=0)),
]


@pytest.mark.parametrize("fmt,bits,akw,rkw", RECALC_FMT_CASES)
def test_recalculate_formats(oracle, mv, fmt, bits, akw, rkw):
    import torch
    w, h, nf = 192, 128, 3
    frames, osup, gsup, osf, gsrc, gsf = _setup(oracle, mv, w, h, bits, nf, fmt, {}, seed=79)
    oan, gan = oracle.Analyse(osup, num_frames=nf, isb=1, **akw), mv.Analyse(gsup, num_frames=nf, isb=1, **akw)
    oold = [oan.frame(osf[n], osf[n + 1] if n + 1 < nf else None) for n in range(nf)]
    gold = gan.run([(gsf[n], gsf[n + 1] if n + 1 < nf else None) for n in range(nf)])
    orc, grc = oracle.Recalculate(osup, oan.ad, **rkw), mv.Recalculate(gsup, gan.ad, **rkw)
    assert grc.blob_size == orc.blob_size
    _same_ad(oracle, grc.ad, orc.ad)
    got = grc.run([(gsf[n], gsf[n + 1] if n + 1 < nf else None, gold[n]) for n in range(nf)])
    torch.cuda.synchronize()
    for n in range(nf):
        want = orc.frame(osf[n], osf[n + 1] if n + 1 < nf else None, oold[n])
        g = got[n].cpu().numpy()
        assert np.array_equal(g, want), (fmt, n, int(np.count_nonzero(g != want)))
