#!/usr/bin/env python3
"""Can two RCCL ranks share ONE GPU on this stack?  (NCCL refuses duplicate devices; this probes RCCL.)  Each rank: init,
one all-reduce on cuda:0, print the result.  Run under `timeout`."""
import os, sys
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
x = torch.full((1024,), float(rank + 1), device="cuda:0")
dist.all_reduce(x)
torch.cuda.synchronize()
print("RANK", rank, "sum", float(x[0]), flush=True)
dist.destroy_process_group()
