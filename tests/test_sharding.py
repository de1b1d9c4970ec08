"""N>1 path on CPU: two gloo processes split a clip into frame ranges (+ tr halo), each runs the pipeline on its shard,
and the gathered result equals the single-process result frame for frame.  (The per-frame work is done by the oracle here --
no GPU in this environment; what is under test is the sharding / halo logic that bench.py and a host application use.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_ranges_partition():
    from mvtools_amd import shard
    for n in (1, 2, 7, 64, 65, 1000):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                s, e = shard.frame_range(n, r, world)
                got += list(range(s, e))
                lo, hi = shard.halo_range(n, r, world, 3)
                if s < e:
                    assert lo == max(0, s - 3) and hi == min(n, e + 3)
            assert got == list(range(n))
    assert shard.ref_index(0, 1, 0, 10) is None and shard.ref_index(0, 1, 1, 10) == 1 and shard.ref_index(9, 2, 1, 10) is None


def _worker(rank, world, port, q):
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import mvoracle as mo
    import pipeline as pl
    from mvtools_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, tr, w, h = 9, 2, 96, 64
    clip = pl.moving_clip(w, h, 8, N, seed=4)           # the "file" every rank can read
    s, e = shard.frame_range(N, rank, world)
    lo, hi = shard.halo_range(N, rank, world, tr)
    sup = mo.Super(w, h, 8)
    supers = {n: sup.frame(clip[n]) for n in range(lo, hi)}    # the rank only touches its shard + halo
    kw = dict(blksize=8, overlap=4)
    ans = {(d, isb): mo.Analyse(sup, isb=isb, delta=d, num_frames=N, **kw) for d in range(1, tr + 1) for isb in (1, 0)}
    dg = mo.Degrain(tr, sup, ans[(1, 1)].ad)
    out = {}
    for n in range(s, e):
        refs, blobs = [], []
        for d in range(1, tr + 1):
            for isb in (1, 0):
                nref = shard.ref_index(n, d, isb, N)
                assert nref is None or lo <= nref < hi, "halo too small"
                refs.append(supers[nref] if nref is not None else None)
                blobs.append(ans[(d, isb)].frame(supers[n], supers[nref] if nref is not None else None))
        out[n] = [int(mo.fnv1a(p)) for p in dg.frame(clip[n], refs, blobs)]
    gathered = [None] * world
    dist.all_gather_object(gathered, out)               # control plane only (test bookkeeping), not a data-path collective
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 29500 + os.getpid() % 2000 + world
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        gathered = q.get(timeout=240)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        merged = {}
        for part in gathered:
            assert not (set(part) & set(merged))
            merged.update(part)
        res[world] = merged
    assert sorted(res[1]) == list(range(9)) == sorted(res[2])
    assert res[1] == res[2]
