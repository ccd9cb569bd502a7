#!/usr/bin/env python3
"""RCCL sanity on a one-GPU box: the collectives bench.py issues (float64 all-reduce SUM / MAX, barrier) in a 1-rank nccl group."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
t = torch.tensor([3.5, 2.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.SUM); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
f = torch.ones(1024, device=dev); dist.all_reduce(f); dist.broadcast(f, src=0)
print("nccl ok", t.tolist(), dist.get_world_size(), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
