"""The VapourSynth filter shell's host logic on a CPU-only machine.

The real plugin (libmvtools_vs.so) runs in the real mini host; a TEST DOUBLE of the device layer (tests/fakedev/mvx_fakedev.c, LD_PRELOAD)
keeps "device" buffers in host memory and answers the three kernel calls of the Super -> Analyse -> Degrain path with the oracle's
functions.  Whatever the shell does with its look-ahead windows, its cache of device frames, its pins, its eviction and its request
threads, the clip that comes out must be the oracle's, byte for byte.  This covers the shell (vsplugin/mvtools_vs.c); the HIP kernels are
covered by the -m gpu suite.  The product never loads the double: it is built here, into a temporary directory."""
import os
import subprocess

import numpy as np
import pytest

import pipeline as pl
from test_vs_shim import HOST, PLUGIN, ROOT, _read_frames, _write_clip, host


def _run(fakedev, args, out, extra, env):
    e = dict(os.environ, LD_PRELOAD=fakedev, MVX_FAKEDEV_STATS="1", MVX_VS_STATS="1")
    e.update(env)
    r = subprocess.run([HOST, PLUGIN] + [str(a) for a in args] + [out] + list(extra), capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    fk = [l for l in r.stderr.splitlines() if l.startswith("mvx_fakedev:")]
    assert len(fk) == 1, r.stderr[-2000:]  # the double was really in front of the library
    return {k: int(v) for k, v in (t.split("=") for t in fk[0].split()[2:])}


def _oracle_degrain(oracle, frames, w, h, bits, radius, blksize, overlap):
    sup = oracle.Super(w, h, bits)
    sf = [sup.frame(f) for f in frames]
    n = len(frames)
    ans = {(d, isb): oracle.Analyse(sup, num_frames=n, isb=isb, delta=d, blksize=blksize, overlap=overlap) for d in range(1, radius + 1) for isb in (1, 0)}
    dg = oracle.Degrain(radius, sup, ans[(1, 1)].ad)
    out = []
    for k in range(n):
        refs, blobs = [], []
        for d in range(1, radius + 1):
            for isb in (1, 0):
                r = k + d if isb else k - d
                ok = 0 <= r < n
                refs.append(sf[r] if ok else None)
                blobs.append(ans[(d, isb)].frame(sf[k], sf[r] if ok else None))
        out.append(dg.frame(frames[k], refs, blobs))
    return out


@pytest.mark.parametrize("bits,radius,threads,env", [
    (8, 1, 1, {"MVX_VS_LOOKAHEAD": "8"}),                                        # five windows and a ragged last one, one request thread
    (16, 3, 8, {"MVX_VS_LOOKAHEAD": "8"}),                                       # six vector clips, eight threads asking for output frames
    (16, 2, 16, {"MVX_VS_LOOKAHEAD": "4", "MVX_VS_LOOKAHEAD_DEPTH": "3"}),       # more threads than a window has frames, three windows ahead
    (8, 2, 8, {"MVX_VS_LOOKAHEAD": "8", "MVX_FAKEDEV_MEM": str(3 << 20)}),       # a "device" so small that cached super frames are evicted all the time
    (16, 1, 8, {"MVX_VS_LOOKAHEAD": "0"}),                                       # the per-frame path with its combining queue
    (16, 3, 8, {"MVX_VS_LOOKAHEAD": "8", "MVX_VS_SUPER_LAZY": "1"}),             # r4 opt-in: super frames whose pixels never leave the device
    (8, 2, 8, {"MVX_VS_LOOKAHEAD": "8", "MVX_VS_SUPER_LAZY": "1", "MVX_FAKEDEV_MEM": str(3 << 20)}),  # ... with evictions: consumers rebuild them from the embedded source
    (16, 1, 4, {"MVX_VS_LOOKAHEAD": "0", "MVX_VS_SUPER_LAZY": "1"}),             # ... on the per-frame path
    (16, 2, 16, {"MVX_VS_LOOKAHEAD": "8", "MVX_VS_MAX_INFLIGHT": "3"}),          # r6 admission gate: sixteen request threads, three output frames admitted at a time
    (8, 1, 12, {"MVX_VS_LOOKAHEAD": "4", "MVX_VS_MAX_INFLIGHT": "1"}),           # ... one at a time (every other thread waits at arInitial)
    (8, 1, 8, {"MVX_VS_LOOKAHEAD": "8", "MVX_VS_MAX_INFLIGHT": "0"}),            # ... switched off
])
def test_shell_reproduces_the_oracle_without_a_gpu(tmp_path, oracle, fakedev, bits, radius, threads, env):
    w, h, n = 160, 96, 37
    frames = pl.moving_clip(w, h, bits, n, seed=5 + radius, noise=3)
    src, out = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    _write_clip(src, frames)
    extra = ["a.blksize=16", "a.overlap=8", "x.threads=%d" % threads, "x.free=1"] + (["x.order=frame"] if threads > 1 else [])
    stats = _run(fakedev, ("run", "degrain%d" % radius, src, w, h, bits, n), out, extra, env)
    got = _read_frames(out, w, h, bits, n)
    want = _oracle_degrain(oracle, frames, w, h, bits, radius, 16, 8)
    for k in range(n):
        for p in range(3):
            assert np.array_equal(got[k][p], want[k][p]), "frame %d plane %d" % (k, p)
    la = int(env["MVX_VS_LOOKAHEAD"])
    assert stats["jobs"] >= n * 2 * radius  # every vector frame was searched (a window may be searched again after its slot was recycled)
    if la:
        windows = -(-n // la)
        assert stats["launches"] >= windows * 2 * radius, stats
        # launches are window-sized on average -- unless the "device" is so small that windows run out of memory: a window that does degrades to the
        # per-frame path (one launch per handful of requests), and how many do depends on how the request threads interleave (r5: this bound was
        # asserted for those configurations too and failed one run in four, with the output bit-identical)
        if "MVX_FAKEDEV_MEM" not in env:
            assert stats["launches"] * (la // 2) <= stats["jobs"] + la * 2 * radius, stats


def test_vector_clip_through_the_shell_equals_the_oracle(tmp_path, oracle, fakedev):
    """mv.Analyse's own output (MVTools_MVAnalysisData + MVTools_vectors of every frame; the mini host builds the delta-1 pair), look-ahead windows of 8, 8 threads"""
    w, h, n, bits = 160, 96, 29, 8
    frames = pl.moving_clip(w, h, bits, n, seed=11, noise=2)
    src, out = str(tmp_path / "in.raw"), str(tmp_path / "vec.raw")
    _write_clip(src, frames)
    _run(fakedev, ("run", "analyse", src, w, h, bits, n), out, ["a.blksize=8", "a.overlap=4", "x.threads=8"], {"MVX_VS_LOOKAHEAD": "8"})
    sup = oracle.Super(w, h, bits)
    sf = [sup.frame(f) for f in frames]
    ans = [oracle.Analyse(sup, num_frames=n, isb=isb, delta=1, blksize=8, overlap=4) for isb in (1, 0)]
    data = open(out, "rb").read()
    pos = 0
    for k in range(n):
        for i, an in enumerate(ans):
            r = k + 1 if i == 0 else k - 1
            blob = an.frame(sf[k], sf[r] if 0 <= r < n else None).tobytes()
            ad = bytes(an.ad)
            assert data[pos:pos + len(ad)] == ad, "analysis data, frame %d clip %d" % (k, i)
            pos += len(ad)
            assert data[pos:pos + len(blob)] == blob, "vectors, frame %d clip %d" % (k, i)
            pos += len(blob)
    assert pos == len(data)


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_shell_under_sanitizers(tmp_path, oracle, sanitizer):
    """The plugin, the mini host and the test double rebuilt with -fsanitize=thread / =address (the "device" buffers are heap blocks here, so an
    out-of-bounds copy or a use of a freed device frame by the shell is a heap error the sanitizer sees): a Degrain3 graph with look-ahead
    windows of 8 and a tiny "device" (constant eviction), 8 request threads, the admission gate admitting three output frames at a time -- no report, and the same
    bytes as the plain build's."""
    rt = subprocess.run(["gcc", "-print-file-name=lib%s.so" % ("tsan" if sanitizer == "thread" else "asan")], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt):
        pytest.skip("gcc has no %s sanitizer runtime here" % sanitizer)
    pkg, odir, inc = os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "include")
    host("list")
    flags = ["-std=gnu11", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer"]
    plug, hst, fake = str(tmp_path / "libmvtools_vs_san.so"), str(tmp_path / "host_san"), str(tmp_path / "libfake_san.so")
    subprocess.check_call(["gcc"] + flags + ["-fPIC", "-shared", "-fvisibility=hidden", os.path.join(pkg, "vsplugin", "mvtools_vs.c"), "-o", plug, "-I" + inc, "-L" + pkg,
                           "-lmvtools_amd", "-Wl,-rpath," + pkg, "-lpthread"])
    subprocess.check_call(["gcc"] + flags + [os.path.join(pkg, "vsplugin", "minihost.c"), "-o", hst, "-ldl", "-lpthread"])
    subprocess.check_call(["gcc"] + flags + ["-shared", "-fPIC", "-I" + inc, "-I" + odir, os.path.join(ROOT, "tests", "fakedev", "mvx_fakedev.c"), "-o", fake, "-L" + odir,
                           "-lmvoracle", "-Wl,-rpath," + odir, "-ldl", "-lpthread"])
    w, h, n, bits = 160, 96, 37, 16
    frames = pl.moving_clip(w, h, bits, n, seed=8, noise=3)
    src, out = str(tmp_path / "in.raw"), str(tmp_path / "out.raw")
    _write_clip(src, frames)
    env = dict(os.environ, LD_PRELOAD=rt + " " + fake, MVX_VS_LOOKAHEAD="8", MVX_FAKEDEV_MEM=str(3 << 20), MVX_VS_STATS="1", MVX_VS_MAX_INFLIGHT="3",  # (r6: eight threads at a gate of three)
               TSAN_OPTIONS="halt_on_error=0", ASAN_OPTIONS="detect_leaks=1:halt_on_error=1")  # (the graph is torn down at the end: what is still allocated then is a leak)
    r = subprocess.run([hst, plug, "run", "degrain3", src, str(w), str(h), str(bits), str(n), out, "a.blksize=16", "a.overlap=8", "x.threads=8", "x.order=frame", "x.free=1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "FREED" in r.stdout and "DONE" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]  # (x.free=1: every filter's free callback ran)
    assert "Sanitizer" not in r.stderr, r.stderr[-4000:]
    assert "permits out" not in r.stderr, r.stderr[-2000:]  # (every admitted request returned its permit before the filters were freed)
    got = _read_frames(out, w, h, bits, n)
    want = _oracle_degrain(oracle, frames, w, h, bits, 3, 16, 8)
    assert all(np.array_equal(got[k][p], want[k][p]) for k in range(n) for p in range(3))
