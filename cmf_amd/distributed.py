"""Multi-GPU execution: one process per GPU, batch sharded on dim 0, weights replicated.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel`` (scatter the batch,
replicate the module each step, gather the per-sample elbos to device 0 and ``.mean()`` them:
``wrapper.py:52-68``, ``non_square_helpers.py:120``).  Samples are independent on this path (no batch
norm in any non-square config), so the MI355X-native form needs no data-path collective at all:
every rank evaluates its shard and the scalar loss is one RCCL all-reduce of (sum, count) -- 8 bytes
over xGMI.  ``torch.distributed`` with backend "nccl" is RCCL on ROCm; CPU tests use "gloo".
"""
import torch
import torch.distributed as dist

__all__ = ["shard_batch", "broadcast_state", "allreduce_mean_elbo"]


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_batch(x, rank=None, world=None):
    """Contiguous, near-even split of dim 0 (what DataParallel's scatter does, wrapper.py:52)."""
    if world is None:
        world = dist.get_world_size() if _active() else 1
        rank = dist.get_rank() if _active() else 0
    n = x.shape[0]
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return x[start:start + base + (1 if rank < extra else 0)]


def broadcast_state(module, src=0):
    """Make parameters AND buffers (the tail's random permutation!) identical on every rank."""
    if not _active():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def allreduce_mean_elbo(elbo):
    """Mean of the per-sample elbos over ALL ranks' shards: all-reduce(sum) of [sum, count]."""
    acc = torch.stack((elbo.sum().double(), torch.tensor(float(elbo.numel()), dtype=torch.float64, device=elbo.device)))
    if _active():
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc[0] / acc[1]
