"""Flat-buffer optimiser step (``cmf_optimizer_step`` / ``cmf_grad_sqnorm``) against torch.optim on the GPU (``-m gpu``)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64, 3, 3), (64,), (7, 5), (1,), (3, 1, 1), (130, 33)]
TORCH = {"sgd": torch.optim.SGD, "adam": torch.optim.Adam, "adamax": torch.optim.Adamax}


def make_params(seed):
    gen = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=gen).cuda()) for s in SHAPES]


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("opt", ["sgd", "adam", "adamax"])
@pytest.mark.parametrize("wd,clip", [(0., None), (0.1, None), (0., 5.), (0.05, 0.5)])
def test_flat_optimizer_matches_torch(opt, wd, clip):
    from cmf_amd.optim import FlatOptimizer
    ref_p, my_p = make_params(1), make_params(1)
    ref = TORCH[opt](ref_p, lr=3e-3, weight_decay=wd)
    mine = FlatOptimizer(my_p, opt=opt, lr=3e-3, weight_decay=wd, max_grad_norm=clip)
    gen = torch.Generator().manual_seed(2)
    for it in range(7):
        ref.zero_grad()
        mine.zero_grad()
        for a, b in zip(ref_p, my_p):
            g = (torch.randn(a.shape, generator=gen) * (3.0 if it % 2 else 0.3)).cuda()
            a.grad = g.clone()
            b.grad.copy_(g)                                         # the views into the flat gradient buffer stay in place
        if clip is not None:
            total = torch.nn.utils.clip_grad_norm_(ref_p, clip)
            assert rel(mine.grad_norm(), total) < 1e-6
        ref.step()
        mine.step()
        for a, b in zip(ref_p, my_p):
            assert rel(b.data, a.data) < 2e-6, (opt, it)
            if clip is not None:
                assert rel(b.grad, a.grad) < 2e-6                   # clipped in place like clip_grad_norm_
    # parameters are views of one buffer; padding slots never move
    assert all(p.data_ptr() == mine.flat.data_ptr() + 4 * o for p, o in zip(my_p, mine.offsets))


@pytest.mark.parametrize("opt", ["adam", "adamax"])
def test_flat_optimizer_state_dict_interchanges_with_torch(opt):
    from cmf_amd.optim import FlatOptimizer
    ref_p, my_p = make_params(3), make_params(3)
    ref = TORCH[opt](ref_p, lr=1e-3)
    gen = torch.Generator().manual_seed(4)
    for _ in range(3):
        for a in ref_p:
            a.grad = torch.randn(a.shape, generator=gen).cuda()
        ref.step()
    for a, b in zip(ref_p, my_p):
        b.data.copy_(a.data)
    mine = FlatOptimizer(my_p, opt=opt, lr=5e-2)
    mine.load_state_dict(copy.deepcopy(ref.state_dict()))           # torch -> flat (lr comes from the checkpoint)
    assert mine.t == 3 and mine.lr == 1e-3
    for _ in range(2):
        for a, b in zip(ref_p, my_p):
            g = torch.randn(a.shape, generator=gen).cuda()
            a.grad = g.clone()
            b.grad = g.clone()                                      # a REPLACED .grad is folded back into the flat buffer
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        assert rel(b.data, a.data) < 2e-6
    fresh = TORCH[opt](ref_p, lr=1e-3)
    fresh.load_state_dict(mine.state_dict())                        # flat -> torch
    second = "exp_avg_sq" if opt == "adam" else "exp_inf"
    for i, a in enumerate(ref_p):
        assert rel(fresh.state[a]["exp_avg"], ref.state[a]["exp_avg"]) < 2e-6
        assert rel(fresh.state[a][second], ref.state[a][second]) < 2e-6
        assert int(fresh.state[a]["step"]) == 5


def test_flat_optimizer_rejects_cpu_parameters():
    from cmf_amd.optim import FlatOptimizer
    with pytest.raises(RuntimeError):
        FlatOptimizer([torch.nn.Parameter(torch.zeros(3))])


def test_flat_optimizer_follows_torch_lr_schedulers():
    """The reference wraps its optimisers in CosineAnnealingLR / LambdaLR (experiment.py:536-552): FlatOptimizer is a
    torch.optim.Optimizer whose lr lives in param_groups[0], so the same schedulers drive the fused step."""
    from cmf_amd.optim import FlatOptimizer
    ref_p, my_p = make_params(5), make_params(5)
    ref = torch.optim.Adam(ref_p, lr=2e-3)
    mine = FlatOptimizer(my_p, opt="adam", lr=2e-3)
    ref_s = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=6, eta_min=0.)
    my_s = torch.optim.lr_scheduler.CosineAnnealingLR(mine, T_max=6, eta_min=0.)
    gen = torch.Generator().manual_seed(6)
    for _ in range(6):
        ref.zero_grad()
        mine.zero_grad()
        for a, b in zip(ref_p, my_p):
            g = torch.randn(a.shape, generator=gen).cuda()
            a.grad = g.clone()
            b.grad.copy_(g)
        ref.step()
        mine.step()
        ref_s.step()
        my_s.step()
        assert abs(mine.lr - ref.param_groups[0]["lr"]) < 1e-12
    for a, b in zip(ref_p, my_p):
        assert rel(b.data, a.data) < 2e-6
