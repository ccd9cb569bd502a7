"""One training step, as ``Trainer._train_batch`` runs it in the reference (``trainer.py:191-224``), for the
one-process-per-GPU layout: closure -> ``loss.backward()`` (the HIP reverse passes, ``densities._ElboFunction``) ->
data-parallel gradient all-reduce over RCCL -> optional global-norm clip -> optimiser step.

``optimizers`` / ``lr_schedulers`` are lists like the reference's (one per objective; the M-flow baseline alternates two).
With ``cmf_amd.optim.FlatOptimizer`` the flat gradient buffer is the all-reduce bucket and clipping happens inside ``step()``;
any ``torch.optim`` optimiser works too (bucketed all-reduce of the ``.grad`` tensors, ``clip_grad_norm_``)."""
import torch

from .distributed import allreduce_gradients
from .optim import FlatOptimizer

__all__ = ["train_batch"]


def train_batch(density, x, epoch, train_metrics, optimizers, lr_schedulers=None, likelihood_introduction_epoch=0,
                max_grad_norm=None):
    """Returns ``{"metrics": train_metrics}`` like ``Trainer._train_batch``; ``x`` is this rank's shard of the batch."""
    n_opt = len(optimizers)
    if epoch < likelihood_introduction_epoch and not epoch % n_opt == 0:        # trainer.py:196-202: skipped objective
        return {"metrics": {"loss": torch.tensor(0.)}}
    density.train()
    opt = optimizers[epoch % n_opt]
    opt.zero_grad()
    metrics = train_metrics(density, x, epoch)
    metrics["loss"].backward()
    if isinstance(opt, FlatOptimizer):
        opt.allreduce_flat(n_local=x.shape[0])
        if max_grad_norm is not None:
            opt.max_grad_norm = max_grad_norm
    else:
        params = [p for group in opt.param_groups for p in group["params"]]
        allreduce_gradients(params, n_local=x.shape[0])
        if max_grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(density.parameters(), max_grad_norm)
    opt.step()
    if lr_schedulers is not None:
        lr_schedulers[epoch % n_opt].step()
    return {"metrics": metrics}
