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
