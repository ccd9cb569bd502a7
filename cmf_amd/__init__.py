"""cmf_amd: MI355X-native implementation of the non-square-flow log-density path of k-flouris/cmf.

Public surface mirrors the reference for this path:
    from cmf_amd import get_config, get_schema, get_density
    density = get_density(get_schema(get_config("mnist", latent_dimension=64)), x_train).cuda().eval()
    density.elbo(x, add_reconstruction=True, add_offdiagonal_metric_reg=True)["elbo"]      # (B, 1)
"""
from .factory import get_density
from .non_square_helpers import get_non_square_parameters, get_non_square_train_metrics
from .schemas import DATA_SHAPES, get_config, get_schema

__all__ = ["get_density", "get_config", "get_schema", "DATA_SHAPES", "get_non_square_train_metrics",
           "get_non_square_parameters"]
