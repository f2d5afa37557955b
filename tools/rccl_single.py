"""RCCL API shapes of the multi-GPU path on a ONE-rank group (diagnostic; GPU box only): the packed float64 all-reduce,
the MAX all-reduce of the timing, barrier(device_ids) and all_gather_object -- the calls bench.py / streaming.py make."""
import os
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
torch.cuda.set_device(dev)
t = torch.arange(1 << 20, dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
m = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier(device_ids=[dev.index])
got = [None]
dist.all_gather_object(got, [(3, np.eye(3), np.ones((1, 3)))])
assert float(t[-1]) == (1 << 20) - 1 and float(m) == 1.5 and got[0][0][0] == 3
dist.destroy_process_group()
print("rccl single-rank API check OK")
