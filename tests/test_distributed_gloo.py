"""N > 1 path on CPU: world_size 2, gloo.  Each rank evaluates its shard (the CPU oracle stands in for the
per-rank HIP evaluation here) and the scalar loss is one all-reduce of (sum, count)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cmf_amd
        from cmf_amd.distributed import allreduce_mean_elbo, broadcast_state, shard_batch
        from conftest import load_golden, golden_model
        from oracle import cmf_oracle as O
        g, meta = load_golden(name)
        _, schema, x_shape, ops, sd = golden_model(meta)
        # rank-dependent construction (random permutation!) must be overwritten by rank 0's state
        torch.manual_seed(100 + rank)
        dens = cmf_amd.get_density(schema, g["x"])
        broadcast_state(dens, src=0)
        flat = torch.cat([v.flatten().double() for v in dens.state_dict().values()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(gathered[0], t) for t in gathered)
        x = g["x"]
        mine = shard_batch(x)
        with torch.no_grad():
            e = O.elbo(sd, ops, mine, add_offdiagonal_metric_reg=True, noise=None if "noise" not in g else shard_batch(g["noise"]))["elbo"]
        mean = allreduce_mean_elbo(e)
        q.put((rank, mine.shape[0], float(mean), same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["c1_sphere", "mini_mnist"])
def test_sharded_mean_elbo_equals_single_process(name):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    g, meta = load_golden(name)
    world, port = 2, 29500 + (os.getpid() % 500) + (7 if name == "mini_mnist" else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sum(r[1] for r in res) == g["x"].shape[0]                    # shards partition the batch (odd B included)
    want = float(g["elbo_0"].double().mean())
    for _, _, mean, same in res:
        assert same, "broadcast_state must make parameters and the permutation buffer identical"
        assert abs(mean - want) < 2e-5 * abs(want)


def test_shard_batch_partitions_unevenly_divisible_batches():
    from cmf_amd.distributed import shard_batch
    x = torch.arange(11).view(11, 1)
    parts = [shard_batch(x, r, 4) for r in range(4)]
    assert [p.shape[0] for p in parts] == [3, 3, 3, 2]
    assert torch.equal(torch.cat(parts), x)
    assert shard_batch(x).shape[0] == 11                                 # no process group: identity


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cmf_amd.distributed import allreduce_gradients
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
        frozen = torch.nn.Parameter(torch.ones(4), requires_grad=False)
        x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10
        mine = x[rank * 4:(rank + 1) * 4]
        net(mine).pow(2).sum().backward()
        if rank == 1:
            net[2].bias.grad = None                  # a parameter without a gradient on one rank
        nb = allreduce_gradients(list(net.parameters()) + [frozen], bucket_bytes=64)     # tiny buckets: several of them
        # plain numpy through the queue: torch tensors travel as file descriptors the parent has to fetch from THIS process,
        # which fails (ConnectionResetError) when the worker has already exited
        q.put((rank, nb, [p.grad.numpy().copy() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


def test_allreduce_gradients_matches_full_batch():
    """world_size 2, gloo: bucketed gradient all-reduce (sum / world) == the two shards' gradients averaged."""
    world, port = 2, 29537
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10
    g = []
    for r in range(2):
        net.zero_grad()
        net(x[r * 4:(r + 1) * 4]).pow(2).sum().backward()
        g.append([p.grad.clone() for p in net.parameters()])
    g[1][3] = torch.zeros_like(g[1][3])              # rank 1 dropped its last-bias gradient
    want = [(a + b) / 2 for a, b in zip(*g)]
    assert res[0][1] == res[1][1] and res[0][1] > 1
    for r in range(2):
        for got, w in zip(res[r][2], want):
            assert torch.allclose(torch.from_numpy(got), w, rtol=1e-6, atol=1e-7)



def _weighted_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cmf_amd.distributed import allreduce_gradients, shard_batch
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))
        x = torch.arange(35, dtype=torch.float32).reshape(7, 5) / 10        # 7 samples over 2 ranks: shards of 4 and 3
        mine = shard_batch(x)
        (-net(mine).mean()).backward()                                       # the per-shard mean loss of the training closure
        allreduce_gradients(list(net.parameters()), n_local=mine.shape[0])
        q.put((rank, mine.shape[0], [p.grad.numpy().copy() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


def test_weighted_gradient_mean_equals_the_global_batch_mean():
    """Unequal shards (7 samples on 2 ranks): sum_r n_r g_r / sum_r n_r is the gradient of the mean over the GATHERED batch,
    which is what the reference's DataParallel + ``elbo.mean()`` differentiates (wrapper.py:52-54, non_square_helpers.py:120);
    the plain mean over ranks is not."""
    world, port = 2, 29561
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_weighted_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [4, 3]
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))
    x = torch.arange(35, dtype=torch.float32).reshape(7, 5) / 10
    (-net(x).mean()).backward()
    for r in range(2):
        for got, p in zip(res[r][2], net.parameters()):
            assert torch.allclose(torch.from_numpy(got), p.grad, rtol=1e-5, atol=1e-7)


def _sum_flat_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cmf_amd.distributed import sum_flat, rs_ag_scratch
        out = {}
        for n in (12, 13, 1, 1000003):                  # a multiple of the world size, ragged, shorter than the world, a large odd one
            g = torch.Generator().manual_seed(100 * n + rank)
            mine = torch.randn(n, generator=g)
            a = sum_flat(mine.clone(), "all_reduce")
            b = sum_flat(mine.clone(), "rs_ag")
            scratch = torch.full((rs_ag_scratch(n) + 5,), float("nan"))       # caller-owned scratch, stale contents
            c = sum_flat(mine.clone(), "rs_ag", scratch)
            out[n] = (a.numpy().copy(), b.numpy().copy(), c.numpy().copy())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])             # 8 = the node the driver scales to
def test_reduce_scatter_all_gather_sums_like_all_reduce(world):
    """``sum_flat``'s two shapes (one all-reduce; reduce-scatter + all-gather over equal, zero-padded slices) give every rank the
    same sums -- the float64 sum of the ranks' buckets to fp32 rounding, and bit-identical to each other at two ranks (one
    addition per element either way)."""
    port = 29571 + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sum_flat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for n in (12, 13, 1, 1000003):
        want = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 * n + r)).double() for r in range(world))
        for r in range(world):
            a, b, c = (torch.from_numpy(t) for t in res[r][n])
            for got in (a, b, c):
                assert got.shape == (n,) and torch.allclose(got.double(), want, rtol=0, atol=4e-7 * world)
            assert torch.equal(b, c) and torch.equal(b, torch.from_numpy(res[0][n][1]))       # every rank holds the same bytes
            if world == 2:
                assert torch.equal(a, b)
