"""Layer-norm lookup behind the TasNet constructors (interface of src/utils/tasnet.py:14-32 of the reference):
'cLN' -> cumulative norm, 'gLN' -> global norm (refused for causal models), 'BN' / 'batch' / 'batch_norm' ->
nn.BatchNorm1d or nn.BatchNorm2d by the `n_dims` keyword."""
import torch.nn as nn

from ..modules.norm import GlobalLayerNorm, CumulativeLayerNorm1d

EPS = 1e-12
_BATCH_NORM_NAMES = ('BN', 'batch', 'batch_norm')
_BATCH_NORM_BY_DIMS = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d}


def choose_layer_norm(name, num_features, causal=False, eps=EPS, **kwargs):
    if name == 'gLN' and causal:
        raise ValueError("Global Layer Normalization is NOT causal.")
    if name == 'gLN':
        return GlobalLayerNorm(num_features, eps=eps)
    if name == 'cLN':
        return CumulativeLayerNorm1d(num_features, eps=eps)
    if name in _BATCH_NORM_NAMES:
        n_dims = kwargs.get('n_dims') or 1
        batch_norm = _BATCH_NORM_BY_DIMS.get(n_dims)
        if batch_norm is None:
            raise NotImplementedError("n_dims is expected 1 or 2, but give {}.".format(n_dims))
        return batch_norm(num_features, eps=eps)
    raise NotImplementedError("Not support {} layer normalization.".format(name))
