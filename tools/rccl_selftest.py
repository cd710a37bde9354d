#!/usr/bin/env python3
"""World-size-1 RCCL smoke: the collectives bench.py / disco_amd.dist use (all_to_all_single with splits,
all_gather_into_tensor, all_reduce, barrier) through backend "nccl" on one GPU.  The 1-GPU test boxes cannot
run more ranks over RCCL; this at least proves the backend initialises and the call shapes are accepted."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(6 * 608, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
w = dist.all_to_all_single(y, x, [6 * 608], [6 * 608], async_op=True)
w.wait()
assert torch.equal(x, y)
c = torch.tensor([3], dtype=torch.int64, device=dev)
o = torch.zeros(1, dtype=torch.int64, device=dev)
dist.all_to_all_single(o, c)
assert int(o) == 3
g = torch.empty(5, dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(g, torch.arange(5, dtype=torch.float64, device=dev))
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl world-1 selftest ok")
