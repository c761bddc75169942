#!/usr/bin/env python3
"""One-rank sanity check of the torch.distributed calls bench.py makes under RCCL ("nccl" backend): barrier, a scalar
all_reduce on the GPU, a gloo side group for python objects.  Run under torchrun on a GPU box:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/nccl_sanity.py"""
import os

import torch
import torch.distributed as dist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world)
obj = dist.new_group(backend="gloo")
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5 + rank], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = [None] * world
dist.all_gather_object(out, {"rank": rank, "x": [1, 2, 3]}, group=obj)
dist.barrier()
dist.destroy_process_group()
print("nccl sanity ok:", float(t.item()), out)
