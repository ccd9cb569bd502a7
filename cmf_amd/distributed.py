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

__all__ = ["shard_batch", "broadcast_state", "allreduce_mean_elbo", "allreduce_gradients", "sum_flat", "REDUCE_SHAPES"]

#: how a flat fp32 bucket is summed over the ranks (``sum_flat``): one ring all-reduce, or the two half-collectives issued
#: directly -- reduce-scatter (every rank receives ONE 1/N slice from each of its N - 1 peers: on the xGMI mesh all 7
#: links of a GPU carry a slice at once) and all-gather.  Same sums; bench.py times both on the node it runs on (N > 1:
#: ``grad_reduce`` object of the line) and its training legs use the faster.
REDUCE_SHAPES = ("all_reduce", "rs_ag")


def _active():
    """A process group exists -- of any size: a one-rank RCCL group still runs its collectives (bench.py --force-group)."""
    return dist.is_available() and dist.is_initialized()


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


def sum_flat(flat, shape="all_reduce", scratch=None):
    """In place: ``flat`` (1-D, contiguous) becomes its sum over all ranks.  ``shape`` "rs_ag": reduce-scatter into this rank's
    1/N slice, then all-gather the slices; the bucket is split in N equal slices, the last one zero-padded in ``scratch``
    (a tensor of at least ``rs_ag_scratch(flat.numel())`` elements, allocated per call when absent).  Every rank must pass
    the same length and shape.  Returns ``flat``."""
    assert shape in REDUCE_SHAPES and flat.dim() == 1 and flat.is_contiguous()
    if not _active():
        return flat
    world = dist.get_world_size()
    if shape == "all_reduce":
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    n = flat.numel()
    c = -(-n // world)                                   # slice length; N c >= n
    need = rs_ag_scratch(n, world)
    if scratch is None:
        scratch = torch.empty(need, dtype=flat.dtype, device=flat.device)
    assert scratch.numel() >= need and scratch.dtype == flat.dtype and scratch.device == flat.device
    mine = scratch[:c]
    if c * world == n:
        dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM)
        dist.all_gather_into_tensor(flat, mine)
    else:
        padded = scratch[c:c + c * world]
        padded[:n].copy_(flat)
        padded[n:].zero_()
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM)
        dist.all_gather_into_tensor(padded, mine)
        flat.copy_(padded[:n])
    return flat


def rs_ag_scratch(n, world=None):
    """Elements of scratch ``sum_flat(..., "rs_ag")`` wants for a bucket of ``n``: one slice, plus a padded copy of the bucket when
    ``n`` is not a multiple of the world size."""
    world = world or (dist.get_world_size() if _active() else 1)
    c = -(-n // world)
    return c + (0 if c * world == n else c * world)


def allreduce_gradients(parameters, bucket_bytes=64 << 20, average=True, n_local=None, shape="all_reduce"):
    """Data-parallel gradient reduction for the one-process-per-GPU trainer: the ``.grad`` tensors are packed
    into a few large flat buckets (the MNIST model's 5.98 M fp32 gradients = 24 MB are ONE bucket) and all-reduced over
    RCCL / xGMI -- the point-to-point links favour few, large collectives over a per-parameter loop.  ``nn.DataParallel``
    in the reference does the equivalent reduce-to-device-0 inside its backward (``wrapper.py:52-68``).  In place; a
    parameter without a gradient on this rank contributes zeros (every rank must call with the same parameter list).
    ``n_local`` = this rank's sample count: gradients are then combined as sum_r n_r g_r / sum_r n_r, the gradient of the
    mean over the GLOBAL batch (``non_square_helpers.py:120`` on the gathered elbos) even when the shards are unequal.
    ``shape``: REDUCE_SHAPES."""
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return 0
    world = dist.get_world_size() if _active() else 1
    weight, total = 1.0, float(world)
    if n_local is not None and average and _active():
        cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=params[0].device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        weight, total = float(n_local), float(cnt.item())
    buckets, cur, size = [], [], 0
    for p in params:
        n = p.numel() * 4
        if cur and size + n > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += n
    buckets.append(cur)
    for group in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in group])
        if _active():
            if weight != 1.0:
                flat *= weight
            sum_flat(flat, shape)
            if average:
                flat /= total
        off = 0
        for p in group:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    return len(buckets)

