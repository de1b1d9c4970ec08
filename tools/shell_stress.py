#!/usr/bin/env python3
"""Developer tool (CPU only): random configurations of the VapourSynth filter shell against the oracle.

The real plugin runs in the real mini host over the test double of the device layer (tests/fakedev/mvx_fakedev.c, see its header): clip size,
bit depth, DegrainN radius, clip length, look-ahead window length and depth, request threads, request order, host cache limit and the size of
the fake "device" (i.e. how hard the frame cache is squeezed) are drawn at random; every output must equal the oracle's byte for byte.

    python tools/shell_stress.py [--seed S] [--runs N] [--sanitize thread|address]

--sanitize rebuilds plugin, mini host and double with that sanitizer (into a temporary directory) and also fails on any sanitizer report.
Round 3: 300 plain runs and 80 ThreadSanitizer runs without a finding.  Round 6 (with the admission gate drawn at random too): see profiles/r6_shell_stress.txt."""
import argparse
import os
import random
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vapoursynth-mvtools_amd")]
import mvoracle as oracle  # noqa: E402
import pipeline as pl  # noqa: E402
from test_vs_shell_cpu import _oracle_degrain  # noqa: E402
from test_vs_shim import HOST, PLUGIN, _read_frames, _write_clip, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--runs", type=int, default=40)
ap.add_argument("--sanitize", choices=["thread", "address"])
args = ap.parse_args()
oracle.lib()
host("list")  # builds the plugin and the mini host if needed
tmp = tempfile.mkdtemp(prefix="shell_stress_")
pkg, odir, inc = os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "include")
flags = ["-std=gnu11", "-O1", "-g"] + (["-fsanitize=" + args.sanitize, "-fno-omit-frame-pointer"] if args.sanitize else [])
fake = os.path.join(tmp, "libfake.so")
subprocess.check_call(["gcc"] + flags + ["-shared", "-fPIC", "-I" + inc, "-I" + odir, os.path.join(ROOT, "tests", "fakedev", "mvx_fakedev.c"), "-o", fake, "-L" + odir, "-lmvoracle",
                       "-Wl,-rpath," + odir, "-ldl", "-lpthread"])
hostbin, plugin, preload = HOST, PLUGIN, fake
if args.sanitize:
    plugin, hostbin = os.path.join(tmp, "libmvtools_vs_san.so"), os.path.join(tmp, "host_san")
    subprocess.check_call(["gcc"] + flags + ["-fPIC", "-shared", "-fvisibility=hidden", os.path.join(pkg, "vsplugin", "mvtools_vs.c"), "-o", plugin, "-I" + inc, "-L" + pkg,
                           "-lmvtools_amd", "-Wl,-rpath," + pkg, "-lpthread"])
    subprocess.check_call(["gcc"] + flags + [os.path.join(pkg, "vsplugin", "minihost.c"), "-o", hostbin, "-ldl", "-lpthread"])
    rt = subprocess.run(["gcc", "-print-file-name=lib%s.so" % ("tsan" if args.sanitize == "thread" else "asan")], capture_output=True, text=True).stdout.strip()
    preload = rt + " " + fake
rnd = random.Random(args.seed)
bad = 0
for it in range(args.runs):
    bits, radius, n = rnd.choice([8, 16]), rnd.choice([1, 2, 3]), rnd.choice([1, 2, 3, 5, 9, 17, 37, 50])
    la, depth, threads = rnd.choice([0, 1, 2, 3, 5, 8, 16, 128]), rnd.choice([1, 2, 3]), rnd.choice([1, 2, 3, 8, 32])
    mem, order, cache = rnd.choice([1 << 20, 3 << 20, 64 << 20]), rnd.choice([True, False]), rnd.choice([None, 4, 16])
    w, h = rnd.choice([(160, 96), (128, 96), (192, 112)])
    gate = rnd.choice([0, 1, 2, 5, 96])  # r6: the admission gate of the consuming filters (MVX_VS_MAX_INFLIGHT)
    cfg = "%dx%d %d-bit Degrain%d %d frames, look-ahead %d x %d, %d threads%s, host cache %s, device %d MiB, gate %d" % (
        w, h, bits, radius, n, la, depth, threads, " (frame order)" if order and threads > 1 else "", cache, mem >> 20, gate)
    frames = pl.moving_clip(w, h, bits, n, seed=1000 * args.seed + it, noise=3)
    src, out = os.path.join(tmp, "in.raw"), os.path.join(tmp, "out.raw")
    _write_clip(src, frames)
    extra = ["a.blksize=16", "a.overlap=8", "x.threads=%d" % threads, "x.free=1"] + (["x.order=frame"] if order and threads > 1 else []) + (["x.cache=%d" % cache] if cache else [])
    env = dict(os.environ, LD_PRELOAD=preload, MVX_VS_LOOKAHEAD=str(la), MVX_VS_LOOKAHEAD_DEPTH=str(depth), MVX_FAKEDEV_MEM=str(mem), MVX_VS_MAX_INFLIGHT=str(gate), TSAN_OPTIONS="halt_on_error=0",
               ASAN_OPTIONS="detect_leaks=1")
    t0 = time.time()
    try:
        r = subprocess.run([hostbin, plugin, "run", "degrain%d" % radius, src, str(w), str(h), str(bits), str(n), out] + extra, capture_output=True, text=True, timeout=600, env=env)
    except subprocess.TimeoutExpired:
        print("HANG ", cfg, flush=True); bad += 1; continue
    verdict = "ok   "
    if r.returncode != 0 or "DONE" not in r.stdout:
        verdict = "FAIL "
    elif "Sanitizer" in r.stderr:
        verdict = "SANIT"
    elif "permits out" in r.stderr:
        verdict = "GATE "
    else:
        got, want = _read_frames(out, w, h, bits, n), _oracle_degrain(oracle, frames, w, h, bits, radius, 16, 8)
        if not all(np.array_equal(got[k][p], want[k][p]) for k in range(n) for p in range(3)):
            verdict = "DIFF "
    print(verdict, cfg, "%.1f s" % (time.time() - t0), flush=True)
    if verdict != "ok   ":
        bad += 1
        print(r.stdout[-500:], r.stderr[-3000:])
print("%d of %d runs bad" % (bad, args.runs))
sys.exit(1 if bad else 0)
