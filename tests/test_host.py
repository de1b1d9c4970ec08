"""CPU tests of the product's host side: the C ABI loads without a GPU, exports every symbol the header declares, and
resolves / validates filter arguments exactly like the oracle's restatement of the reference's Create functions."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(mv):
    hdr = open(os.path.join(ROOT, "include", "mvtools_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(mvx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 24
    lib = mv.lib()
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing


def test_wire_struct_sizes(mv):
    assert C.sizeof(mv.AnalysisData) == 84          # MVAnalysisData, MVAnalysisData.h:83-134
    assert C.sizeof(mv.AnalyseJob) == 7 * 8 + 8
    assert C.sizeof(mv.DegrainJob) == (3 + 36 + 12 + 3) * 8


SUPER_ARGS = [dict(), dict(pel=1), dict(pel=4, sharp=0, rfilter=4), dict(hpad=8, vpad=4), dict(levels=3), dict(chroma=0),
              dict(pel=3), dict(sharp=3), dict(rfilter=5), dict(sharp=-1)]


@pytest.mark.parametrize("kw", SUPER_ARGS)
@pytest.mark.parametrize("fmt", [(640, 360, 8, (1, 1)), (1920, 1080, 8, (1, 1)), (3840, 2160, 16, (1, 1)), (720, 486, 10, (1, 0)), (133, 77, 8, (0, 0))])
def test_super_create_matches_oracle(oracle, mv, fmt, kw):
    w, h, bits, sub = fmt
    try:
        o = oracle.Super(w, h, bits, subsampling=sub, **kw)
        oerr = None
    except oracle.OracleError as e:
        o, oerr = None, str(e)
    try:
        g = mv.Super(w, h, bits, subsampling=sub, **kw)
        gerr = None
    except mv.MvtoolsError as e:
        g, gerr = None, str(e)
    assert gerr == oerr
    if o is not None:
        assert (g.info.super_width, g.info.super_height, g.info.levels, g.info.modeYUV, g.info.pel, g.info.hpad, g.info.vpad) == \
               (o.s.superWidth, o.s.superHeight, o.s.levels, o.s.modeYUV, o.s.pel, o.s.hpad, o.s.vpad)


ANALYSE_ARGS = [dict(), dict(blksize=16, overlap=8), dict(blksize=32, overlap=16, isb=1, delta=3), dict(blksize=4, overlap=2), dict(blksize=8, blksizev=4),
                dict(truemotion=0), dict(lambda_=777, lsad=900, pnew=30, pzero=10, pglobal=5, plevel=2), dict(levels=2), dict(levels=-2),
                dict(search=3, searchparam=5), dict(search=1, searchparam=-4), dict(chroma=0), dict(badsad=2000, badrange=-8), dict(pelsearch=4),
                # invalid
                dict(blksize=12), dict(overlap=6), dict(search=9), dict(search_coarse=-1), dict(plevel=3), dict(pnew=300), dict(pzero=-1),
                dict(pglobal=257), dict(levels=99), dict(blksize=8, overlap=3), dict(delta=-100), dict(blksize=16, blksizev=2, dct=0),
                dict(divide=3)]


@pytest.mark.parametrize("kw", ANALYSE_ARGS)
@pytest.mark.parametrize("fmt", [(640, 360, 8), (1920, 1080, 16)])
def test_analyse_create_matches_oracle(oracle, mv, fmt, kw):
    w, h, bits = fmt
    osup, gsup = oracle.Super(w, h, bits), mv.Super(w, h, bits)
    try:
        o = oracle.Analyse(osup, num_frames=50, **kw)
        oerr = None
    except oracle.OracleError as e:
        o, oerr = None, str(e)
    try:
        g = mv.Analyse(gsup, num_frames=50, **kw)
        gerr = None
    except mv.MvtoolsError as e:
        g, gerr = None, str(e)
    assert (gerr is None) == (oerr is None), (gerr, oerr)
    if oerr is not None:
        assert gerr == oerr
    else:
        assert g.blob_size == o.blob_size
        for k, _ in oracle.AnalysisData._fields_:
            if k not in ("nMagicKey", "nVersion", "nCPUFlags"):
                assert getattr(g.ad, k) == getattr(o.ad, k), k


def test_unimplemented_modes_fail_loudly(mv):
    sup = mv.Super(640, 360, 8)
    with pytest.raises(mv.MvtoolsError):
        mv.Analyse(sup, dct=1)  # FFTW DCT cost modes 1..4
    rc = mv.Recalculate
    with pytest.raises(mv.MvtoolsError):
        rc(sup, mv.Analyse(sup).ad, dct=3)


@pytest.mark.parametrize("kw", [dict(), dict(thsad=200, thsadc=100, plane=0), dict(limit=5, limitc=7), dict(thscd1=300, thscd2=90),
                                dict(plane=5), dict(limit=300), dict(limitc=-2), dict(thscd1=999999), dict(thsad=1 << 40)])
def test_degrain_create_matches_oracle(oracle, mv, kw):
    osup, gsup = oracle.Super(640, 360, 8), mv.Super(640, 360, 8)
    oan, gan = oracle.Analyse(osup, blksize=8, overlap=4), mv.Analyse(gsup, blksize=8, overlap=4)
    try:
        oracle.Degrain(2, osup, oan.ad, **kw)
        oerr = None
    except oracle.OracleError as e:
        oerr = str(e)
    try:
        mv.Degrain(2, gsup, gan.ad, [640, 320, 320], **kw)
        gerr = None
    except mv.MvtoolsError as e:
        gerr = str(e)
    assert (gerr is None) == (oerr is None), (gerr, oerr)


def test_scale_thscd_matches_oracle(oracle, mv):
    for (bits, blk, chroma) in [(8, 8, 1), (16, 16, 1), (10, 32, 0), (8, 4, 1)]:
        osup, gsup = oracle.Super(640, 360, bits), mv.Super(640, 360, bits)
        oan, gan = oracle.Analyse(osup, blksize=blk, chroma=chroma), mv.Analyse(gsup, blksize=blk, chroma=chroma)
        for (t1, t2) in [(400, 130), (100, 255), (16320, 0)]:
            a1, a2 = C.c_int64(t1), C.c_int(t2)
            oracle.lib().mvo_scale_thscd(C.byref(a1), C.byref(a2), C.byref(oan.d.ad))
            b1, b2 = C.c_int64(t1), C.c_int32(t2)
            mv.lib().mvx_scale_thscd(C.byref(b1), C.byref(b2), C.byref(gan.ad))
            assert (a1.value, a2.value) == (b1.value, b2.value)


def test_no_cpu_fallback(mv):
    """without a GPU the compute entry points must fail, not silently run somewhere else"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sup = mv.Super(64, 48, 8)
    with pytest.raises(mv.MvtoolsError):
        sup.alloc(1)


def test_public_header_is_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/mvtools_amd.h must compile as C99 and as C++ on its own (no torch / HIP types)"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "mvtools_amd.h"\nint main(void) { mvx_super_args a; (void)a; return (int)sizeof(mvx_analysis_data) == 84 ? 0 : 1; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        exe = tmp_path / ("t_" + cc)
        r = subprocess.run([cc, std, "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-x", "c" if cc == "gcc" else "c++", str(src), "-o", str(exe)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert subprocess.run([str(exe)]).returncode == 0


def test_shadow_layout_is_checked_before_the_device_is_touched(mv):
    """with shadow planes Super writes, and Analyse reads, more than height x pitch bytes behind a plane pointer; a frame
    that was not allocated with that room is refused on the host (no GPU needed: the check precedes every device call)."""
    import torch
    sup = mv.Super(64, 48, 16)
    assert sup.shadow and max(sup.slots) > 1
    H, W = sup.info.plane_height, sup.pitch  # (alloc() itself refuses to run without a device: the layout it makes, by hand)
    good = [torch.zeros(sup.shadow_stride[p] * sup.slots[p], dtype=torch.uint8)[:H[p] * W[p]].view(H[p], W[p]) for p in range(sup.nplanes)]
    sup.check_room([good])
    plain = [torch.zeros(sup.info.plane_height[p], sup.pitch[p], dtype=torch.uint8) for p in range(sup.nplanes)]
    with pytest.raises(ValueError, match="Super.alloc"):
        sup.check_room([plain])
    with pytest.raises(ValueError, match="Super.alloc"):
        sup._shadows([plain])
    an = mv.Analyse(sup, num_frames=4, blksize=16)
    with pytest.raises(ValueError, match="Analyse input"):
        an.run([(good, plain)], blobs=[torch.zeros(an.blob_size, dtype=torch.uint8)])
    sup8 = mv.Super(64, 48, 8, subsampling=(0, 0))  # 4:4:4 8-bit: no shadow data at all, any tensor of the right pitch will do
    if not sup8.shadow:
        sup8.check_room([[torch.zeros(sup8.info.plane_height[p], sup8.pitch[p], dtype=torch.uint8) for p in range(sup8.nplanes)]])
