"""Loss closures of the non-square trainer: our counterpart of ``cmf/non_square_helpers.py``.

``get_non_square_train_metrics(config)`` returns ``(train_metrics, likelihood_introduction_epoch,
early_stopping_start_epoch)`` exactly like the reference (:31-135); ``train_metrics(density, x,
epoch)`` calls ``density.elbo`` with the reference's kwarg names and weights, so a reference-style
trainer drives the HIP path unchanged.
"""
import numpy as np

__all__ = ["get_non_square_parameters", "get_non_square_train_metrics"]


def get_non_square_parameters(module, m_flow):
    """Optimiser parameter groups (reference :8-28): one group, or reconstruction / likelihood groups
    of the M-flow head, found by walking ``module`` / ``density`` / ``prior`` children."""
    if not m_flow:
        return [module.parameters()]
    node = module
    while type(node).__name__ != "ManifoldFlowHeadDensity":
        for child in ("module", "density", "prior"):
            if child in node._modules:
                node = node._modules[child]
                break
        else:
            raise RuntimeError(f"Module {node} has no prior")
    return node.separate_parameters()


def _schedule(config):
    """Epoch -> (likelihood_weight, add_reconstruction): warm-up ramp of :35-58 (np.interp between the
    warm-up bounds), with the M-flow convention that two epochs make one (alternating objectives)."""
    n_obj = 2 if config["m_flow"] else 1
    if config["likelihood_warmup"]:
        bounds = [n_obj * config["likelihood_warmup_start"], n_obj * config["likelihood_warmup_end"]]
        intro, early = bounds
    else:
        bounds, intro, early = None, 0, 0

    def at(epoch):
        on = (epoch + 1) % n_obj == 0
        if bounds is not None:
            w = float(np.interp(epoch, bounds, [0, 1])) if on else 0
        else:
            w = float(on)
        return w, epoch % n_obj == 0

    return at, intro, early


def get_non_square_train_metrics(config):
    at, intro, early = _schedule(config)

    def train_metrics(density, x, epoch):
        w, rec = at(epoch)
        loss = -density.elbo(x, likelihood_wt=w, add_reconstruction=rec)["elbo"].mean()
        return {"loss": loss}

    def train_metrics_l1_diagonal(density, x, epoch):
        w, rec = at(epoch)
        loss = -density.elbo(x, likelihood_wt=w * config["elbo_regularization_param"],
                             metric_wt=w * config["metric_regularization_param"], add_reconstruction=rec,
                             add_diagonal_metric_reg=rec, add_offdiagonal_metric_reg=False)["elbo"]
        return {"loss": loss.mean()}

    def train_metrics_l1_offdiagonal(density, x, epoch):
        w, rec = at(epoch)
        loss = -density.elbo(x, likelihood_wt=w * config["elbo_regularization_param"],
                             metric_wt=w * config["metric_regularization_param"], add_reconstruction=rec,
                             add_diagonal_metric_reg=False, add_offdiagonal_metric_reg=rec)["elbo"]
        return {"loss": loss.mean()}

    if config["g_kk_loss"]:
        assert config["g_ij_loss"] == False, "Cannot have both diagonal and offdiagonal terms in l1 yet.  Exiting..."
        return train_metrics_l1_diagonal, intro, early
    if config["g_ij_loss"]:
        assert config["latent_dimension"] != 1, "There is no offdiagonal for 1d latent. Exiting..."
        return train_metrics_l1_offdiagonal, intro, early
    return train_metrics, intro, early
