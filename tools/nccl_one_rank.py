"""developer tool: RCCL sanity on ONE rank (init, barrier, all_reduce, all_gather_object -- the calls bench.py makes at N > 1): python tools/nccl_one_rank.py"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
torch.cuda.set_device(0)
t=torch.tensor([1.5],device="cuda",dtype=torch.float64)
dist.barrier(); dist.all_reduce(t, op=dist.ReduceOp.MAX); torch.cuda.synchronize()
g=[None]; dist.all_gather_object(g, {"rank":0})
print("nccl world 1 ok", t.item(), g)
dist.destroy_process_group()
