"""Measurement of the drop-in path: the VapourSynth filter shell (libmvtools_vs.so) driven by the mini host with T
request threads on a 4K YUV420P16 clip -- mv.Super -> mv.Analyse x 6 -> mv.Degrain3 -- timed wall-clock (host copies, PCIe and
the per-frame shell work included), and compared bit for bit with the same graph evaluated through the batched C ABI (the
Python binding), which the parity suite ties to the oracle.   usage (GPU box):  python tools/vs_4k_run.py [frames] [threads]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "tests")]
import pipeline as pl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 36
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w, h, bits, tr = 3840, 2160, 16, 3
tmp = os.environ.get("TMPDIR", "/tmp")
src, out = os.path.join(tmp, "vs4k_in.raw"), os.path.join(tmp, "vs4k_out.raw")
t0 = time.time()
per_frame = (w * h + 2 * (w // 2) * (h // 2)) * 2
verify = os.environ.get("VS_NOVERIFY") is None
if verify or not (os.path.exists(src) and os.path.getsize(src) == per_frame * N):  # (VS_NOVERIFY=1: a parameter sweep reuses the clip file)
    frames = pl.moving_clip(w, h, bits, N, seed=3, noise=2)
    with open(src, "wb") as f:
        for fr in frames:
            for p in fr:
                f.write(np.ascontiguousarray(p).tobytes())
    print("clip: %d frames %dx%d P%d written in %.1f s" % (N, w, h, bits, time.time() - t0), flush=True)

host, plugin = os.path.join(ROOT, "vapoursynth-mvtools_amd", "mvx_vs_host"), os.path.join(ROOT, "vapoursynth-mvtools_amd", "libmvtools_vs.so")
env = dict(os.environ, MVX_VS_STATS="1", MVX_HOST_TIMES="1")
t0 = time.time()
r = subprocess.run([host, plugin, "run", "degrain3", src, str(w), str(h), str(bits), str(N), out, "a.blksize=16", "a.overlap=8", "x.threads=%d" % T] + (["x.order=frame"] if os.environ.get("VS_ORDER", "frame") == "frame" else []) + (["x.cache=%s" % os.environ["VS_CACHE"]] if os.environ.get("VS_CACHE") else []),
                   capture_output=True, text=True, env=env)
dt = time.time() - t0
if os.environ.get("MVX_VS_TRACE"):  # the window events of the look-ahead (one line each), kept whole
    trace_lines = [ln for ln in r.stderr.splitlines() if "trace" in ln]
    open(os.path.join(ROOT, "gpurun_out", "vs_trace_events.txt"), "w").write("\n".join(trace_lines) + "\n")
    r.stderr = "\n".join(ln for ln in r.stderr.splitlines() if "trace" not in ln)
print(r.stdout.strip()[-200:], r.stderr.strip()[-700:], flush=True)
assert r.returncode == 0 and "DONE" in r.stdout
print("shell: %d output frames, %d request threads: %.1f s wall = %.2f fps (reads the clip file, uploads, PCIe both ways, writes the result file)" % (N, T, dt, N / dt), flush=True)
import re  # noqa: E402
m = re.search(r"output clip \(frame order\) ([0-9.]+) s", r.stderr)
if m:
    print("steady state: %d frames requested in frame order in %s s = %.1f fps (graph construction, which runs the first windows, and the result file excluded)" % (N, m.group(1), N / float(m.group(1))), flush=True)
marks = [(int(a), float(b)) for a, b in re.findall(r"minihost: (\d+) requests done at ([0-9.]+) s", r.stderr)]
if os.environ.get("VS_MARKS"):
    print("progress (requests done, seconds): " + " ".join("%d:%.2f" % m for m in marks), flush=True)
if len(marks) >= 4:  # the second half of the run: device arenas and host buffers are being recycled by then
    (n0, t0_), (n1, t1_) = marks[len(marks) // 2 - 1], marks[-1]
    print("second half of the run: frames %d .. %d in %.2f s = %.1f fps" % (n0, n1, t1_ - t0_, (n1 - n0) / (t1_ - t0_)), flush=True)
if not verify:
    sys.exit(0)

import torch  # noqa: E402
import mvtools_amd as mv  # noqa: E402
sup = mv.Super(w, h, bits)
gsrc = [mv.frame_to_device(f) for f in frames]
sf = sup.build(gsrc)
ans = {(d, isb): mv.Analyse(sup, num_frames=N, isb=isb, delta=d, blksize=16, overlap=8) for d in range(1, tr + 1) for isb in (1, 0)}
blobs = {}
for (d, isb), a in ans.items():
    jobs = []
    for n in range(N):
        nref = n + d if isb else n - d
        jobs.append((sf[n], sf[nref] if 0 <= nref < N else None))
    blobs[(d, isb)] = a.run(jobs)
dg = mv.Degrain(tr, sup, ans[(1, 1)].ad, [p.stride(0) for p in gsrc[0]])
djobs = []
for n in range(N):
    refs, bl = [], []
    for d in range(1, tr + 1):
        for isb in (1, 0):
            nref = n + d if isb else n - d
            refs.append(sf[nref] if 0 <= nref < N else None)
            bl.append(blobs[(d, isb)][n])
    djobs.append((gsrc[n], refs, bl))
res = dg.run(djobs)
torch.cuda.synchronize()
got = np.fromfile(out, dtype=np.uint16)
per = w * h + 2 * (w // 2) * (h // 2)
assert got.size == per * N
bad = 0
for n in range(N):
    o = 0
    for p in range(3):
        pw, ph = (w, h) if p == 0 else (w // 2, h // 2)
        want = mv.plane_to_numpy(res[n][p], pw, np.uint16)
        g = got[n * per + o:n * per + o + pw * ph].reshape(ph, pw)
        o += pw * ph
        if not np.array_equal(g, want):
            bad += 1
print("shell output == batched C-ABI output on all %d frames x 3 planes: %s" % (N, "yes" if bad == 0 else "NO (%d planes differ)" % bad))
sys.exit(1 if bad else 0)
