"""developer tool: run the pipeline once (for rocprofv3)."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", "vapoursynth-mvtools_amd"), os.path.dirname(__file__), os.path.join(os.path.dirname(__file__), "..")]
import torch, mvtools_amd as mv
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pipe = bench.Pipeline(mv, torch, cfg, B, torch.device("cuda"), 1)
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    pipe.step()
torch.cuda.synchronize()
print("done")
