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


def _worker(rank, world, port, q, N=9, tr=2):
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import mvoracle as mo
    import pipeline as pl
    from mvtools_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 96, 64
    clip = pl.moving_clip(w, h, 8, N, seed=4)           # the "file" every rank can read
    plan = shard.RankPlan(N, rank, world, tr)           # the same planner bench.py's Pipeline builds its job tables from
    lo, hi = plan.held
    assert (lo, hi) == shard.halo_range(N, rank, world, tr) and plan.out == shard.frame_range(N, rank, world)
    sup = mo.Super(w, h, 8)
    supers = [sup.frame(clip[n]) for n in range(lo, hi)]       # the rank only touches its shard + halo (local indices)
    kw = dict(blksize=8, overlap=4)
    ans = {(d, isb): mo.Analyse(sup, isb=isb, delta=d, num_frames=N, **kw) for d, isb in plan.clips}
    dg = mo.Degrain(tr, sup, ans[(1, 1)].ad)
    if lo == hi:  # more ranks than frames: this rank owns nothing (it still takes part in the gather)
        supers = []
    blobs = {key: [ans[key].frame(supers[n], supers[nref] if nref is not None else None) for n, nref in pairs]
             for key, pairs in plan.searches().items()}
    out = {}
    for (n, refs, i), gn in zip(plan.degrains(), plan.outputs()):
        out[gn] = [int(mo.fnv1a(p)) for p in dg.frame(clip[gn], [supers[r] if r is not None else None for r in refs], [blobs[key][i] for key in plan.clips])]
    gathered = [None] * world
    dist.all_gather_object(gathered, out)               # control plane only (test bookkeeping), not a data-path collective
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


def _run_worlds(worlds, N, tr):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for world in worlds:
        q = ctx.Queue()
        port = 29500 + (os.getpid() * 7 + world * 13 + N) % 2000
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, N, tr)) for r in range(world)]
        for p in procs:
            p.start()
        gathered = q.get(timeout=280)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        merged = {}
        for part in gathered:
            assert not (set(part) & set(merged))
            merged.update(part)
        res[world] = merged
    return res


@pytest.mark.timeout(600)
def test_uneven_sharding_radius_6_worlds_3_and_8():
    """r5: a clip whose length does not divide by the world size (11 frames over 3 and over 8 ranks: shares of 4+4+3 and of 2+2+2+1+1+1+1+1), temporal
    radius 6 (Degrain6: the halo is longer than any share, every rank holds most of the clip, references beyond the clip ends are missing exactly as in
    the single-process run): the gathered result equals the world-1 result frame for frame."""
    N, tr = 11, 6
    res = _run_worlds((1, 3, 8), N, tr)
    assert sorted(res[1]) == list(range(N)) == sorted(res[3]) == sorted(res[8])
    assert res[1] == res[3] == res[8]


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


def test_rank_plan_matches_bench_layout():
    """bench.py's clip: world*B output frames + a tr-frame lead-in / lead-out; every rank gets B frames and a full halo"""
    from mvtools_amd import shard
    B, tr = 5, 3
    for world in (1, 2, 8):
        seen = []
        for r in range(world):
            p = shard.RankPlan(world * B + 2 * tr, r, world, tr, first_out=tr, last_out=world * B + tr)
            assert len(p.outputs()) == B and p.held == (p.out[0] - tr, p.out[1] + tr)
            seen += list(p.outputs())
            s = p.searches()
            assert list(s) == [(d, isb) for d in range(1, tr + 1) for isb in (1, 0)]
            for (d, isb), pairs in s.items():
                assert all(nref == (n + d if isb else n - d) and 0 <= nref < B + 2 * tr for n, nref in pairs)
            for n, refs, i in p.degrains():
                assert n == tr + i and refs == [n + 1, n - 1, n + 2, n - 2, n + 3, n - 3]
        assert seen == list(range(tr, world * B + tr))
    # a plain clip (no lead-in): references beyond the clip ends are None, exactly like the filters (MVAnalyse.c:120-129)
    p = shard.RankPlan(4, 0, 1, 2)
    assert p.searches()[(2, 0)][:3] == [(0, None), (1, None), (2, 0)] and p.degrains()[3][1] == [None, 2, None, 1]


def test_bench_rank_plumbing(monkeypatch, capsys):
    """bench.py --gpus N: WORLD_SIZE must agree with --gpus, and without a launcher it starts N ranks itself -- or refuses loudly"""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    assert bench.world_from_env(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}) == (1, 1, 2)
    assert bench.world_from_env(1, {}) == (0, 0, 1)
    with pytest.raises(SystemExit):
        bench.world_from_env(8, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    with pytest.raises(SystemExit):
        bench.world_from_env(1, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    import torch
    if torch.cuda.device_count() < 2:  # (here: no GPU at all) asking for two ranks must fail, not silently run one
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode != 0 and "--gpus 2" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_rank_plans_on_one_gpu_match_the_single_rank_run(world):
    """The N>1 path on the GPU as far as one GPU allows: the ranks of a world of 2 (3) run one after the other on the same device
    -- each with bench.py's own Pipeline, its RankPlan share of ONE clip and the tr-frame halo it runs mv.Super on itself -- and the
    union of their outputs (vector blobs of every clip and the Degrain planes) must equal the world-1 run frame for frame.  No
    collective is involved: SURVEY.md 8(e)."""
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vapoursynth-mvtools_amd")]
    import bench
    import mvtools_amd as mv
    from mvtools_amd import shard
    w, h, bits, tr, total = 320, 192, 16, 2, 11
    cfg = (w, h, bits, tr, dict(blksize=16, overlap=8), dict(pel=2), 0, "sharding test clip")
    N = total + 2 * tr
    dev = torch.device("cuda", 0)
    clip = bench.synth_clip_device(torch, w, h, bits, N, seed=5, device=dev)
    torch.cuda.synchronize()

    def run(rank, nranks):
        plan = shard.RankPlan(N, rank, nranks, tr, first_out=tr, last_out=N - tr)
        outs = list(plan.outputs())
        p = bench.Pipeline(mv, torch, cfg, len(outs), dev, seed=0, src=clip[plan.held[0]:plan.held[1]], plan=plan)
        p.step()
        torch.cuda.synchronize()
        res = {}
        for i, n in enumerate(outs):
            res[n] = ([pl_.clone() for pl_ in p.out[i]], {k: p.blobs[k][i].clone() for k in plan.clips})
        return res, plan

    whole, _ = run(0, 1)
    union = {}
    for r in range(world):
        part, plan = run(r, world)
        assert plan.held[0] == max(0, plan.out[0] - tr) and plan.held[1] == min(N, plan.out[1] + tr)
        assert not (set(part) & set(union)), "ranks must own disjoint frame ranges"
        union.update(part)
    assert sorted(union) == sorted(whole) == list(range(tr, N - tr))
    for n in whole:
        for pi, (a, b) in enumerate(zip(whole[n][0], union[n][0])):
            rb = (w >> (1 if pi else 0)) * 2  # (the pitch padding of the output planes is never written)
            assert torch.equal(a[:, :rb], b[:, :rb]), "Degrain output of frame %d differs between the world-1 and the world-%d run" % (n, world)
        for k in whole[n][1]:
            assert torch.equal(whole[n][1][k], union[n][1][k]), "vectors of frame %d, clip %s differ" % (n, k)


@pytest.mark.gpu
def test_ranks_generate_identical_shared_frames_and_the_shard_check_sees_a_difference():
    """bench.py --gpus N as far as one GPU allows (r5): every rank generates the frames it holds from the clip's seed and the GLOBAL frame index, so
    the frames two neighbouring ranks share (the halo) are identical bytes; bench.shard_payload / shard_verdict (what main() runs over
    all_gather_object at N > 1) confirm it, check one output frame on either side of the shard boundary against the oracle, and report a
    corrupted halo frame."""
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "vapoursynth-mvtools_amd")]
    import bench
    import mvtools_amd as mv
    from mvtools_amd import shard
    w, h, bits, tr, B, world = 320, 192, 16, 2, 5, 2
    cfg = (w, h, bits, tr, dict(blksize=16, overlap=8), dict(pel=2), 0, "shard check test clip")
    N = world * B + 2 * tr
    dev = torch.device("cuda", 0)
    whole = bench.synth_clip_device(torch, w, h, bits, N, seed=1000, device=dev)
    pipes, plans = [], []
    for r in range(world):
        plan = shard.RankPlan(N, r, world, tr, first_out=tr, last_out=N - tr)
        p = bench.Pipeline(mv, torch, cfg, B, dev, seed=1000, plan=plan)      # generates plan.held from the global index, like main()
        for k, n in enumerate(range(*plan.held)):
            for a, b in zip(p.src[k], whole[n]):
                assert torch.equal(a, b), "rank %d: frame %d differs from the whole clip's" % (r, n)
        p.step()
        torch.cuda.synchronize()
        pipes.append(p)
        plans.append(plan)
    gathered = [bench.shard_payload(mv, torch, cfg, pipes[r], plans[r], r, 4) for r in range(world)]
    v = bench.shard_verdict(gathered)
    assert v["shared_source_frames_compared"] == 2 * tr and v["shared_source_frames_identical"] and v["output_ranges_partition_the_job"], v
    assert [b["identical"] for b in v["boundary_frames_vs_oracle"]] == [True, True] and "errors" not in v, v
    assert [b["global_frame"] for b in v["boundary_frames_vs_oracle"]] == [tr + B - 1, tr + B]
    pipes[1].src[0][0][3, 5] += 1                                             # rank 1's copy of a frame rank 0 holds too
    torch.cuda.synchronize()
    gathered[1] = bench.shard_payload(mv, torch, cfg, pipes[1], plans[1], 1, 4, oracle=False)
    v = bench.shard_verdict(gathered)
    assert not v["shared_source_frames_identical"] and "frame %d" % plans[1].held[0] in v["mismatches"][0], v
