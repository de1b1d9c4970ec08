"""Shared test plumbing: the same Super -> Analyse -> Degrain/Compensate pipeline driven through the oracle (CPU) and
through the product's C ABI (GPU), on identical inputs."""
import numpy as np


def moving_clip(width, height, bits, nframes, seed=7, noise=2, motion=(3, -1), sub=(1, 1)):
    """Textured 4:2:0 (or other subsampling) clip: band-limited texture translating by `motion` px/frame, a rectangle
    moving the other way, plus +-noise LSB (8-bit scale) uniform noise.  SURVEY.md 8(d)."""
    rng = np.random.default_rng(seed)
    scale = 1 if bits == 8 else (1 << (bits - 8))
    pm = (1 << bits) - 1
    big_h, big_w = height + 64 + 8 * nframes, width + 64 + 8 * nframes
    yy, xx = np.mgrid[0:big_h, 0:big_w].astype(np.float64)
    tex = (40 * np.sin(xx * 0.21 + yy * 0.07) + 30 * np.sin(xx * 0.05 - yy * 0.13) + 20 * np.sin(xx * 0.33 + 1.3) * np.cos(yy * 0.27)
           + 25 * (((xx.astype(int) // 8) + (yy.astype(int) // 8)) & 1) + 120)
    texc = [(20 * np.sin(xx * 0.11 + yy * 0.05 + k) + 128) for k in (0.3, 1.7)]
    frames = []
    for f in range(nframes):
        ox, oy = 32 + 4 * nframes + motion[0] * f, 32 + 4 * nframes + motion[1] * f
        planes = []
        for p in range(3):
            sx, sy = (1 << sub[0], 1 << sub[1]) if p else (1, 1)
            w, h = width // sx, height // sy
            base = tex if p == 0 else texc[p - 1]
            img = base[oy:oy + height:sy, ox:ox + width:sx][:h, :w].copy()
            # independently moving rectangle (-2,+2)/frame carrying its own texture (a different part of the field)
            rx, ry = (width // 2 - 2 * f) // sx, (height // 3 + 2 * f) // sy
            rw, rh = (width // 4) // sx, (height // 4) // sy
            y0r, x0r = max(ry, 0), max(rx, 0)
            patch = base[8:8 + (ry + rh - y0r) * sy:sy, 8:8 + (rx + rw - x0r) * sx:sx]
            ph_, pw_ = min(patch.shape[0], h - y0r), min(patch.shape[1], w - x0r)
            if ph_ > 0 and pw_ > 0:
                img[y0r:y0r + ph_, x0r:x0r + pw_] = patch[:ph_, :pw_] * 0.8 + (35 if p == 0 else 10)
            img = img + rng.integers(-noise, noise + 1, img.shape)
            v = np.clip(np.rint(img * scale), 0, pm)
            planes.append(v.astype(np.uint8 if bits == 8 else np.uint16))
        frames.append(planes)
    return frames


def crop(plane, width):
    return plane[:, :width]


def defined_equal(oracle_sup, oracle_frame, gpu_frame_np):
    """compare only the defined rectangles of a super frame; returns list of mismatching regions"""
    bad = []
    for (p, lv, k, y0, x0, h, w) in oracle_sup.defined_regions():
        a = oracle_frame[p][y0:y0 + h, x0:x0 + w]
        b = gpu_frame_np[p][y0:y0 + h, x0:x0 + w]
        if not np.array_equal(a, b):
            ys, xs = np.nonzero(a != b)
            bad.append((p, lv, k, int(len(ys)), int(ys[0]), int(xs[0]), int(a[ys[0], xs[0]]), int(b[ys[0], xs[0]])))
    return bad


def blob_vectors(blob, ad, level=0):
    """numpy view (x, y, sad) of one level of a MVTools_vectors blob"""
    b = np.asarray(blob, dtype=np.uint8)
    off = 8
    nWB = (ad.nBlkSizeX - ad.nOverlapX) * ad.nBlkX + ad.nOverlapX
    nHB = (ad.nBlkSizeY - ad.nOverlapY) * ad.nBlkY + ad.nOverlapY
    for i in range(ad.nLvCount - 1, -1, -1):
        bx = ((nWB >> i) - ad.nOverlapX) // (ad.nBlkSizeX - ad.nOverlapX)
        by = ((nHB >> i) - ad.nOverlapY) // (ad.nBlkSizeY - ad.nOverlapY)
        n = bx * by
        if i == level:
            rec = b[off + 4: off + 4 + n * 16]
            xy = rec.view(np.int32).reshape(n, 4)[:, :2]
            sad = rec.view(np.int64).reshape(n, 2)[:, 1]
            return xy[:, 0].reshape(by, bx), xy[:, 1].reshape(by, bx), sad.reshape(by, bx)
        off += 4 + n * 16
    raise ValueError(level)
