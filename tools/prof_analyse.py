"""developer tool: per-phase cycle breakdown of the search kernel (needs a library built with MVX_PROFILE=1)."""
import ctypes as C, sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", "vapoursynth-mvtools_amd"), os.path.dirname(__file__), os.path.join(os.path.dirname(__file__), "..")]
import torch, mvtools_amd as mv
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pipe = bench.Pipeline(mv, torch, cfg, B, torch.device("cuda"), 1)
pipe.step(); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
assert mv.lib().mvx_debug_prof(out) == 0
v = list(out)
nb = max(v[3], 1)
print("blocks", v[3], "rounds(passes)", v[8], "passes/block %.2f" % (v[8] / nb))
print("per block cycles: prologue %.0f  search %.0f  epilogue %.0f  | total kernel cycles %d  (%.0f / block)" % (v[0] / nb, v[1] / nb, v[2] / nb, v[9], v[9] / nb))
print("state machine per block: prelude %.0f  round-call %.0f  post %.0f | prologue: stage %.0f prefetch %.0f window %.0f rest %.0f" % (v[10]/nb, v[11]/nb, v[12]/nb, v[13]/nb, v[14]/nb, v[15]/nb, (v[0]-v[13]-v[14]-v[15])/nb))
print("per pass cycles: eval(load+sad) %.0f  cost+argmin %.0f | fast passes with a window miss: %d of %d" % (v[4] / max(v[8], 1), v[6] / max(v[8], 1), v[5], v[8]))
