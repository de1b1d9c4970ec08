"""Deterministic synthetic clips shared by the tests, bench.py and smoke() (no reference code involved).

`survey_clip` reproduces the generator SURVEY.md 8(c)/8(d) describes (LCG x <- x*1664525 + 1013904223, seed 12345,
top 16 bits): 8x8 checker translating by 2 px/frame plus LSB noise; the three blob hashes recorded in SURVEY.md were
produced on exactly these clips.
"""
import numpy as np


class LCG:
    A = 1664525
    C = 1013904223

    def __init__(self, seed=12345):
        self.state = np.uint64(seed)

    def take(self, n):
        """next n outputs (state >> 16 after each step), vectorised jump-ahead."""
        a = np.full(n, self.A, dtype=np.uint64)
        with np.errstate(over="ignore"):
            ak = np.cumprod(a)                       # a^1 .. a^n  (mod 2^64)
            s = np.cumsum(np.concatenate([[np.uint64(1)], ak[:-1]]))  # sum_{j<k} a^j, k=1..n
            st = (ak * self.state + np.uint64(self.C) * s) & np.uint64(0xFFFFFFFF)
        self.state = st[-1]
        return (st >> np.uint64(16)).astype(np.uint32)


def _cdiv(a, b):
    """C integer division (truncation toward zero)."""
    return np.trunc(a / b).astype(np.int64)


def survey_clip(width, height, bits=8, nframes=2, half_flat=False, seed=12345):
    """frames[f] = [Y, U, V] 4:2:0 planes; frame f is the checker shifted by 2*f (luma) px for f in {0,1}
    (f>=2 keeps shifting by 2 px/frame)."""
    rng = LCG(seed)
    frames = []
    for f in range(nframes):
        planes = []
        for p in range(3):
            w, h = (width // 2, height // 2) if p else (width, height)
            cell = 4 if p else 8
            shift = (f * (1 if p else 2))
            x = np.arange(w)[None, :]
            y = np.arange(h)[:, None]
            xs = x - shift
            checker = ((_cdiv(xs, cell) + y // cell) & 1)
            r = rng.take(w * h).reshape(h, w).astype(np.int64)
            if bits == 8:
                if half_flat:
                    v = 128 + np.where(x < w // 2, 60 * checker, 0) + (r & 7)
                else:
                    v = 128 + 60 * checker + (r & 3)
                planes.append((v & 0xFF).astype(np.uint8))
            else:
                v = 257 * (128 + 60 * checker) + (r & 1023)
                planes.append((v & 0xFFFF).astype(np.uint16))
        frames.append(planes)
    return frames
