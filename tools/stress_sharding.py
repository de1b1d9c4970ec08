"""developer / evidence tool: tests/test_sharding.py's GPU case (rank plans of one clip on one GPU == the world-1 run) N times in ONE
process, worlds 2 and 3 alternating -- the situation in which round 4's stream-ordering race showed (a later parametrisation, kernels
already loaded, the host enqueues faster).  python tools/stress_sharding.py [N=30]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "tests")]
import test_sharding as ts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
t0 = time.time()
for i in range(n):
    for world in (2, 3):
        try:
            ts.test_rank_plans_on_one_gpu_match_the_single_rank_run.__wrapped__(world) if hasattr(ts.test_rank_plans_on_one_gpu_match_the_single_rank_run, "__wrapped__") else ts.test_rank_plans_on_one_gpu_match_the_single_rank_run(world)
        except AssertionError as e:
            bad += 1
            print("iteration %d world %d: FAILED: %s" % (i, world, e))
print("stress_sharding: %d iterations x worlds (2, 3): %d failures, %.1f s" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
