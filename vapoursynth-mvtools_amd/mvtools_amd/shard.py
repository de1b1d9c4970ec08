"""Frame-range sharding across GPUs (one process per GPU).

Output frame n of mv.DegrainN depends only on input frames n-tr .. n+tr (MVDegrains.cpp:92-109, MVAnalyse.c:84-98) and no
state carries between frames, so a clip is split into contiguous frame ranges, one per rank, each extended by a tr-frame
halo that the rank loads and runs mv.Super on itself.  There is no data-path collective."""


def frame_range(num_frames, rank, world):
    """[start, stop) of the output frames owned by `rank`: contiguous, disjoint, covering, sizes differ by at most 1."""
    base, extra = divmod(num_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def halo_range(num_frames, rank, world, tr):
    """[lo, hi) of the INPUT frames rank needs: its output range widened by tr on both sides, clipped to the clip."""
    s, e = frame_range(num_frames, rank, world)
    if s == e:
        return s, e
    return max(0, s - tr), min(num_frames, e + tr)


def ref_index(n, delta, isb, num_frames):
    """reference frame of vector clip (delta, isb) at frame n, or None when it falls outside the clip (MVAnalyse.c:120-129,187-221)"""
    nref = n + delta if isb else n - delta
    return nref if 0 <= nref < num_frames else None


class RankPlan:
    """What one rank does for a clip of `num_frames` input frames and vector clips (delta 1..tr, isb 1/0): the output
    frames it owns, the input frames it must hold, and -- in indices LOCAL to that held range -- the (current, reference)
    pairs of every search and the per-frame reference lists of DegrainN.  bench.py, the CPU sharding test and a host
    application all derive their job tables from this one class."""

    def __init__(self, num_frames, rank, world, tr, first_out=0, last_out=None):
        """Output frames are the clip's frames [first_out, last_out) (default: all); they are dealt to the ranks in
        contiguous ranges.  A bench clip that carries its own tr-frame lead-in / lead-out passes first_out = tr,
        last_out = num_frames - tr so that every reference exists."""
        last_out = num_frames if last_out is None else last_out
        self.num_frames, self.rank, self.world, self.tr = num_frames, rank, world, tr
        s, e = frame_range(last_out - first_out, rank, world)
        self.out = (first_out + s, first_out + e)
        self.held = (max(0, self.out[0] - tr), min(num_frames, self.out[1] + tr)) if s < e else (self.out[0], self.out[0])
        self.clips = [(d, isb) for d in range(1, tr + 1) for isb in (1, 0)]  # DegrainN's argument order: bw1, fw1, bw2, fw2, ...

    def local(self, n):
        return None if n is None else n - self.held[0]

    def outputs(self):
        return range(self.out[0], self.out[1])

    def searches(self):
        """{(delta, isb): [(local current frame, local reference frame or None), ...]} over the rank's output frames"""
        res = {}
        for d, isb in self.clips:
            res[(d, isb)] = []
            for n in self.outputs():
                nref = ref_index(n, d, isb, self.num_frames)
                assert nref is None or self.held[0] <= nref < self.held[1], "halo too small"
                res[(d, isb)].append((self.local(n), self.local(nref)))
        return res

    def degrains(self):
        """[(local frame, [local reference frame or None per vector clip], index of the frame's blob in each clip's list)]"""
        res = []
        for i, n in enumerate(self.outputs()):
            res.append((self.local(n), [self.local(ref_index(n, d, isb, self.num_frames)) for d, isb in self.clips], i))
        return res
