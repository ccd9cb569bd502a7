"""Data-parallel training step on the HIP path (``-m gpu``): two processes share the one GPU of the test box and talk over gloo
(RCCL refuses two ranks on one device; on a node each rank has its own GPU and the backend is "nccl").  Covers
``training.train_batch`` -> ``loss.backward()`` -> ``FlatOptimizer.allreduce_flat`` -> fused Adam across ranks."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, reduce_shape):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cmf_amd
        from cmf_amd.optim import FlatOptimizer
        from cmf_amd.training import train_batch
        from test_gpu_parity import build
        torch.cuda.set_device(0)
        g, meta, cfg, dens = build("mini_mnist")
        dens = dens.module.density                      # no dequantisation noise: the two layouts must see the same inputs
        cfg = dict(cfg, g_ij_loss=True, g_kk_loss=False)
        train_metrics, _, _ = cmf_amd.get_non_square_train_metrics(cfg)
        shape = tuple(g["x"].shape[1:])
        x = torch.randint(0, 256, (8, *shape), generator=torch.Generator().manual_seed(3)).float().cuda()
        if world == 1:
            shard = x
        else:
            shard = x[rank * 4:(rank + 1) * 4]
        opt = FlatOptimizer(dens.parameters(), opt="adam", lr=1e-3, reduce_shape=reduce_shape)
        out = train_batch(dens, shard.clone(), 10_000, train_metrics, [opt])
        q.put((rank, float(out["metrics"]["loss"].detach()), opt.grad.cpu().numpy().copy(), opt.flat.cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


def _run(world, port, reduce_shape="all_reduce"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, reduce_shape)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    return res


def test_two_rank_training_step_equals_the_full_batch_step():
    import numpy as np
    port = 29650 + os.getpid() % 200
    two = _run(2, port)
    one = _run(1, port + 1)
    g1, p1 = one[0][2], one[0][3]
    for rank in range(2):
        # averaged shard gradients == the gradient of the mean loss over the full batch; identical parameters after the step
        assert np.abs(two[rank][2] - g1).max() <= 1e-5 * np.abs(g1).max()
        assert np.abs(two[rank][3] - p1).max() <= 1e-6 * np.abs(p1).max() + 1e-7
    assert abs(0.5 * (two[0][1] + two[1][1]) - one[0][1]) <= 1e-5 * abs(one[0][1])
    # the bucket summed as reduce-scatter + all-gather (FlatOptimizer.reduce_shape): at two ranks the same single addition per element
    # -- the reduced gradients may differ from the all-reduce's only through the backward's own run-to-run atomics
    rs = _run(2, port + 2, "rs_ag")
    for rank in range(2):
        assert np.array_equal(rs[rank][2], rs[0][2]) and np.array_equal(rs[rank][3], rs[0][3])      # the ranks stay replicas
        assert np.abs(rs[rank][2] - g1).max() <= 1e-5 * np.abs(g1).max()
        assert np.abs(rs[rank][3] - p1).max() <= 1e-6 * np.abs(p1).max() + 1e-7
