"""Checkpoint files in the reference's format.

The reference trainer writes ``<logdir>/checkpoints/<tag>.pt`` = ``torch.save`` of a dict with the keys
``epoch, iteration, module_state_dict, opt_state_dicts, best_valid_loss, num_bad_valid_epochs,
lr_scheduler_state_dicts`` (``trainer.py:362-375``), through a temporary file + ``os.replace`` so that a checkpoint on
disk is always complete (``writer.py:105-122``), and reads it back with ``torch.load(map_location=device)``
(``writer.py:117-118``, ``trainer.py:377-400``, including the older single ``opt_state_dict`` /
``lr_scheduler_state_dict`` spelling).  ``cmf_amd`` densities use the reference's ``state_dict()`` key schema
(``tests/test_host_logic.py`` checks it against the reference's own dump), so a file written by either side loads on the
other.  Host-side only: no kernels involved.
"""
import math
import os

import torch

KEYS = ("epoch", "iteration", "module_state_dict", "opt_state_dicts", "best_valid_loss", "num_bad_valid_epochs",
        "lr_scheduler_state_dicts")


def checkpoint_path(logdir, tag):
    """``writer.py:120-126``: ``<logdir>/checkpoints/<tag>.pt``."""
    return os.path.join(logdir, "checkpoints", f"{tag}.pt")


def write_checkpoint(logdir, tag, density, optimizers=(), lr_schedulers=(), epoch=0, iteration=0,
                     best_valid_loss=math.inf, num_bad_valid_epochs=0):
    """Same dict, same path, same atomic replace as ``Trainer._save_checkpoint`` / ``Writer.write_checkpoint``."""
    data = {
        "epoch": int(epoch),
        "iteration": int(iteration),
        "module_state_dict": density.state_dict(),
        "opt_state_dicts": [o.state_dict() for o in optimizers],
        "best_valid_loss": best_valid_loss,
        "num_bad_valid_epochs": int(num_bad_valid_epochs),
        "lr_scheduler_state_dicts": [s.state_dict() for s in lr_schedulers],
    }
    path = checkpoint_path(logdir, tag)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = os.path.join(os.path.dirname(path), os.path.basename(path) + ".tmp")
    torch.save(data, tmp)
    os.replace(tmp, path)                       # atomic: a checkpoint on disk is always whole
    return path


def _match_prefix(state, want_keys):
    """A reference model trained on several GPUs carries nn.DataParallel's ``module.`` prefix (``factory.py:76-81``); accept
    a file written with or without it for a model built the other way."""
    have = next(iter(state), "")
    want = next(iter(want_keys), "")
    if have.startswith("module.") and not want.startswith("module."):
        return {k[len("module."):]: v for k, v in state.items()}
    if want.startswith("module.") and not have.startswith("module."):
        return {"module." + k: v for k, v in state.items()}
    return state


def load_checkpoint(logdir, tag, density, optimizers=(), lr_schedulers=(), device="cpu"):
    """``Trainer._load_checkpoint``: restores the module (strictly), optimisers and schedulers (new or old key spelling)
    and returns the whole dict (``epoch``, ``iteration``, ``best_valid_loss``, ``num_bad_valid_epochs`` are the caller's)."""
    ckpt = torch.load(checkpoint_path(logdir, tag), map_location=device, weights_only=False)
    density.load_state_dict(_match_prefix(ckpt["module_state_dict"], density.state_dict().keys()))

    def load_list(key, old_key, objects):
        if not objects:
            return
        if key in ckpt:
            for obj, sd in zip(objects, ckpt[key]):
                obj.load_state_dict(sd)
        else:                                   # trainer.py:394-396: older files hold a single state dict
            objects[0].load_state_dict(ckpt[old_key])

    load_list("opt_state_dicts", "opt_state_dict", list(optimizers))
    load_list("lr_scheduler_state_dicts", "lr_scheduler_state_dict", list(lr_schedulers))
    return ckpt
