#!/usr/bin/env python3
"""RCCL sanity on a one-GPU box: the collectives bench.py issues (float64 all-reduce SUM / MAX, barrier, broadcast, and the gradient
bucket's two reduction shapes: all-reduce, reduce-scatter + all-gather) in a 1-rank nccl group."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
t = torch.tensor([3.5, 2.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.SUM); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
f = torch.ones(1024, device=dev); dist.all_reduce(f); dist.broadcast(f, src=0)
from cmf_amd.distributed import sum_flat, rs_ag_scratch
import bench
b = torch.arange(1003, dtype=torch.float32, device=dev)             # cmf_amd.distributed.sum_flat: one rank, the bucket comes back unchanged
for shape in ("all_reduce", "rs_ag"):
    assert torch.equal(sum_flat(b.clone(), shape), b) and torch.equal(sum_flat(b.clone(), shape, torch.empty(rs_ag_scratch(1003), device=dev)), b)
gr = bench.grad_reduce_leg(1 << 20, 0, 1, dev, iters=3, warmup=1)   # the N > 1 line's leg, as bench.py runs it
assert gr["used"] in ("all_reduce", "rs_ag") and gr["all_reduce"] > 0 and gr["rs_ag"] > 0, gr
print("nccl ok", t.tolist(), dist.get_world_size(), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
