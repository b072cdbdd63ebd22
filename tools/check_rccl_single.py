#!/usr/bin/env python
"""RCCL smoke on one rank: process-group init with device_id, broadcast, async all_reduce on a side stream (the calls the
data-parallel path makes); the N > 1 behaviour is covered on CPU by the gloo test."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(1 << 20, dtype=torch.bfloat16, device="cuda")
dist.broadcast(x, src=0)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    w = dist.all_reduce(x, async_op=True)
w.wait()
torch.cuda.current_stream().wait_stream(side)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", float(x[:4].float().sum()))
dist.destroy_process_group()
