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
