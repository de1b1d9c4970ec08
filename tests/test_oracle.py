"""CPU tests of the oracle (test infrastructure): pinning against the reference's recorded outputs, against the
reference's own object code (oracle/_ref, when built) and against committed golden fixtures."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import pipeline as pl
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "..", "oracle", "_ref", "libmvref.so")

# vector-blob hashes recorded in SURVEY.md 8(c) from the reference itself (scalar and SSE2/AVX2 builds agree)
SURVEY_KATS = [(128, 96, 8, "6d02e770"), (128, 96, 16, "b899c7de"), (640, 360, 8, "9ee5c88e")]


@pytest.mark.parametrize("w,h,bits,want", SURVEY_KATS)
def test_survey_known_answers(oracle, w, h, bits, want):
    fr = synth.survey_clip(w, h, bits)
    sup = oracle.Super(w, h, bits)
    s0, s1 = sup.frame(fr[0]), sup.frame(fr[1])
    an = oracle.Analyse(sup, blksize=8, overlap=4, opt=0)
    blob = an.frame(s0, s1)
    assert "%08x" % oracle.fnv1a(blob) == want
    x, y, _ = pl.blob_vectors(blob, an.ad, 0)
    assert int(x[0, 1]) == 4 and int(y[0, 1]) == 0  # the synthetic 2-px shift, in half-pel units


def test_geometry_table(oracle):
    """SURVEY.md section 8 config table (computed there from the reference formulas)."""
    want = {  # (w, h, bits, blk, ov, pel): (superW, superH, levels, nBlkX, nBlkY, lvCount, blob bytes)
        (640, 360, 8, 8, 0, 1): (672, 978, 8, 80, 45, 6, 76224),
        (1920, 1080, 8, 8, 4, 2): (1952, 5822, 10, 479, 269, 8, 2738792),
        (3840, 2160, 16, 16, 8, 2): (3872, 11254, 11, 479, 269, 8, 2738792),
        (1920, 1080, 8, 8, 0, 2): (1952, 5822, 10, 240, 135, 8, 688920),
        (7680, 4320, 16, 32, 16, 2): (7712, 22086, 12, 479, 269, 8, 2738792),
    }
    for (w, h, bits, blk, ov, pel), exp in want.items():
        s = oracle.Super(w, h, bits, pel=pel)
        a = oracle.Analyse(s, blksize=blk, overlap=ov)
        got = (s.s.superWidth, s.s.superHeight, s.s.levels, a.ad.nBlkX, a.ad.nBlkY, a.ad.nLvCount, a.blob_size)
        assert got == exp, ((w, h, bits, blk, ov, pel), got)


def _ref():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (needs /root/reference; built by __graft_entry__.build())")
    r = C.CDLL(REF_SO)
    vp, ss = C.c_void_p, C.c_ssize_t
    r.ref_sad.restype = C.c_uint
    r.ref_sad.argtypes = [C.c_int] * 4 + [vp, ss, vp, ss]
    r.ref_satd.restype = C.c_uint
    r.ref_satd.argtypes = [C.c_int] * 3 + [vp, ss, vp, ss]
    r.ref_over_init.argtypes = [vp] + [C.c_int] * 4
    r.ref_overlaps.argtypes = [C.c_int] * 4 + [vp, ss, vp, ss, vp, ss]
    r.ref_to_pixels.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    r.ref_refine_avx2.argtypes = [C.c_int, vp, vp, ss, ss, ss]
    r.ref_average2_avx2.argtypes = [vp, vp, vp, ss, ss, ss]
    r.ref_copy.argtypes = [C.c_int] * 3 + [vp, ss, vp, ss]
    return r


SIZES = [(2, 2), (2, 4), (4, 2), (4, 4), (4, 8), (8, 1), (8, 2), (8, 4), (8, 8), (8, 16), (16, 1), (16, 2), (16, 4), (16, 8), (16, 16),
         (16, 32), (32, 8), (32, 16), (32, 32), (32, 64), (64, 16), (64, 32), (64, 64), (64, 128), (128, 32), (128, 64), (128, 128)]


def test_sad_satd_against_reference_objects(oracle):
    """SADFunctions.cpp / SADFunctions_AVX2.cpp compiled from /root/reference vs the oracle's restatement.
    The source block is contiguous (pitch == width), as in the reference (PlaneOfBlocks.cpp:1066-1069): its 8-bit SIMD
    kernels ignore the source pitch (SADFunctions.cpp:63)."""
    r, L = _ref(), oracle.lib()
    rng = np.random.default_rng(1)
    for bits in (8, 16):
        dt, hi = (np.uint8, 256) if bits == 8 else (np.uint16, 65536)
        for (w, h) in SIZES:
            for trial in range(3):
                a = rng.integers(0, hi, (h, w)).astype(dt)
                b = rng.integers(0, hi, (h + 3, w + 19)).astype(dt)
                if trial == 2:
                    a[:], b[:] = 0, hi - 1
                o = L.mvo_sad(w, h, bits, a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0])
                for avx in (0, 1):
                    assert r.ref_sad(w, h, bits, avx, a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0]) == o
        for (w, h) in [(4, 4), (8, 4), (8, 8), (16, 8), (16, 16), (32, 16), (32, 32), (64, 32), (64, 64), (128, 64), (128, 128)]:
            for trial in range(4):
                a = rng.integers(0, hi, (h, w + 5)).astype(dt)
                b = rng.integers(0, hi, (h + 3, w + 19)).astype(dt)
                if trial == 3:
                    a[:, ::2], b[:, 1::2] = 0, hi - 1
                assert (r.ref_satd(w, h, bits, a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0]) ==
                        L.mvo_satd(w, h, bits, a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0]))


def test_overlap_against_reference_objects(oracle):
    """Overlap.cpp / Overlap_AVX2.cpp: window tables (float + cosf), overlaps accumulate, ToPixels."""
    r, L = _ref(), oracle.lib()
    rng = np.random.default_rng(2)
    for (nx, ny, ox, oy) in [(8, 8, 4, 4), (16, 16, 8, 8), (32, 32, 16, 16), (4, 4, 2, 2), (8, 8, 2, 2), (16, 8, 4, 2), (16, 16, 4, 8),
                             (64, 64, 32, 32), (128, 128, 64, 64), (8, 4, 4, 2), (2, 2, 1, 1), (16, 16, 0, 8), (16, 16, 8, 0), (32, 16, 8, 4)]:
        a = np.zeros(9 * nx * ny, np.int16)
        b = np.zeros(9 * nx * ny, np.int16)
        r.ref_over_init(a.ctypes.data, nx, ny, ox, oy)
        L.mvo_over_init(b.ctypes.data, nx, ny, ox, oy)
        assert np.array_equal(a, b), (nx, ny, ox, oy)
    for bits in (8, 16):
        dt, hi, acc = (np.uint8, 256, np.uint16) if bits == 8 else (np.uint16, 65536, np.uint32)
        for (w, h) in [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (16, 8), (2, 2), (64, 64)]:
            win = rng.integers(0, 2049, (h, w)).astype(np.int16)
            src = rng.integers(0, hi, (h, w + 3)).astype(dt)
            d0 = rng.integers(0, 1000, (h, w + 8)).astype(acc)
            d1, d2 = d0.copy(), d0.copy()
            L.mvo_overlaps(w, h, bits, d1.ctypes.data, d1.strides[0], src.ctypes.data, src.strides[0], win.ctypes.data, w)
            r.ref_overlaps(w, h, bits, 0, d2.ctypes.data, d2.strides[0], src.ctypes.data, src.strides[0], win.ctypes.data, w)
            assert np.array_equal(d1, d2)
            if w >= 4:
                d3 = d0.copy()
                r.ref_overlaps(w, h, bits, 1, d3.ctypes.data, d3.strides[0], src.ctypes.data, src.strides[0], win.ctypes.data, w)
                assert np.array_equal(d1, d3)
        accs = rng.integers(0, (hi * 40), (20, 40)).astype(acc)
        o1 = np.zeros((20, 40), dt)
        o2 = np.zeros((20, 40), dt)
        L.mvo_to_pixels(bits, o1.ctypes.data, o1.strides[0], accs.ctypes.data, accs.strides[0], 40, 20)
        r.ref_to_pixels(bits, o2.ctypes.data, o2.strides[0], accs.ctypes.data, accs.strides[0], 40, 20)
        assert np.array_equal(o1, o2)


def test_refine_kernels_against_reference_objects(oracle):
    """MVFrame_AVX2.cpp (8-bit bilinear / Wiener half-pel kernels, Average2) vs the oracle's scalar restatement of
    MVFrame.cpp:508-572,1019-1111,1180-1197.  The AVX2 kernels overshoot the width into the stride gap,
    so only the defined width is compared."""
    r, L = _ref(), oracle.lib()
    rng = np.random.default_rng(3)
    for (w, h) in [(64, 40), (96, 33), (160, 50), (37, 21)]:
        pitch = (w + 63) // 64 * 64 + 64
        src = rng.integers(0, 256, (h + 8, pitch)).astype(np.uint8)
        for kind in (0, 1, 2, 5, 6):
            d1 = np.zeros((h + 8, pitch), np.uint8)
            d2 = np.zeros((h + 8, pitch), np.uint8)
            L.mvo_refine_plane(kind, 8, d1.ctypes.data, src.ctypes.data, pitch, w, h)
            assert r.ref_refine_avx2(kind, d2.ctypes.data, src.ctypes.data, pitch, w, h) == 0
            assert np.array_equal(d1[:h, :w], d2[:h, :w]), (w, h, kind)
        b = rng.integers(0, 256, (h + 8, pitch)).astype(np.uint8)
        d1 = np.zeros((h + 8, pitch), np.uint8)
        d2 = np.zeros((h + 8, pitch), np.uint8)
        L.mvo_average2(8, d1.ctypes.data, src.ctypes.data, b.ctypes.data, pitch, w, h)
        r.ref_average2_avx2(d2.ctypes.data, src.ctypes.data, b.ctypes.data, pitch, w, h)
        assert np.array_equal(d1[:h, :w], d2[:h, :w])


def _golden_cases():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["name"])
def test_golden_fixtures(oracle, case):
    """Committed vectors (tests/golden/golden.json, written by tests/golden/make_golden.py from the oracle AFTER it had
    matched the reference's recorded hashes): guards the oracle against regressions on configurations the three SURVEY
    hashes do not reach (pel 1/4, other reduce / sharp filters, Degrain, Compensate)."""
    import make_golden
    got = make_golden.run_case(oracle, case["params"])
    assert got == case["expect"], case["name"]


def test_mask_upsizer_against_reference_object(oracle):
    """mv.BlockFPS's bilinear mask upsizer (SimpleResize.cpp:62-121): the oracle's restatement against the reference's own AVX2
    object code (SimpleResize_AVX2.cpp), fed with the same offset / weight tables (InitTables itself needs <VSHelper.h> and is
    not built; the table restatement is the one the GPU path is checked against)."""
    r = _ref()
    vp = C.c_void_p
    r.ref_simple_resize_u8_avx2.argtypes = [vp, C.c_int, vp, C.c_int] + [C.c_int] * 4 + [vp] * 4
    L = oracle.lib()
    L.mvo_simple_resize_u8.argtypes = [vp, C.c_int, vp, C.c_int] + [C.c_int] * 4
    L.mvo_resize_tables.argtypes = [vp, vp, C.c_int, C.c_int]
    rng = np.random.default_rng(3)
    for (sw, sh, dw, dh) in [(60, 34, 480, 272), (31, 18, 248, 144), (240, 135, 1920, 1080), (16, 9, 64, 36), (40, 30, 43, 33), (12, 7, 200, 113)]:
        src = rng.integers(0, 256, (sh + 1, sw + 8), dtype=np.uint8)  # (+1 row: the last output rows read offset + 1; +8: vpgatherdd over-read)
        src[::3, ::5] = 255
        src[1::4, 2::7] = 0
        want = np.zeros((dh, dw + 8), np.uint8)
        got = np.zeros_like(want)
        L.mvo_simple_resize_u8(got.ctypes.data, got.shape[1], src.ctypes.data, src.shape[1], dw, dh, sw, sh)
        vo, vw, ho, hw = (np.zeros(dh, np.int32), np.zeros(dh, np.int32), np.zeros(dw + 8, np.int32), np.zeros(dw + 8, np.int32))
        L.mvo_resize_tables(ho.ctypes.data, hw.ctypes.data, dw, sw)
        L.mvo_resize_tables(vo.ctypes.data, vw.ctypes.data, dh, sh)
        hw[:dw] = (hw[:dw] << 16) | (16384 - hw[:dw])  # the AVX2 kernel's packed form of the weights (simpleInit, SimpleResize.cpp:152-155)
        r.ref_simple_resize_u8_avx2(want.ctypes.data, want.shape[1], src.ctypes.data, src.shape[1], dw, dh, sw, sh, vo.ctypes.data, vw.ctypes.data, ho.ctypes.data, hw.ctypes.data)
        assert np.array_equal(got[:, :dw], want[:, :dw]), (sw, sh, dw, dh)


def test_search_statistics_tool_runs():
    """tools/search_stats.py (the oracle built with -DMVO_STATS, counters only): runs on a small clip and its percentages are sane."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "search_stats.py"), "256", "144", "8", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    m = re.search(r"level 0: (\d+) blocks", r.stdout)
    assert m and int(m.group(1)) == 2 * 31 * 17  # two searches, (256 / 8 - 1) x (144 / 8 - 1) blocks of 16 with overlap 8
    ends = re.search(r"predictor phase ends on[^:]*: (.*)", r.stdout).group(1)
    assert abs(sum(float(v) for v in re.findall(r"([0-9.]+) %", ends)) - 100.0) < 0.5
