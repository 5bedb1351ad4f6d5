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


def frame_padding(n_samples, kernel_size, stride):
    """(left, right) zero samples that put a whole number of hops behind the first analysis window: the shortfall
    (stride - (T - K) % stride) % stride, its smaller half in front (src/models/conv_tasnet.py:137-141)."""
    shortfall = -(n_samples - kernel_size) % stride
    left = shortfall // 2
    return left, shortfall - left


def waveform_rows(input):
    """A mixture batch as rows of samples for the encoder: (B, 1, T) stays, (B, 1, mics, T) becomes (B, mics, T).
    Returns (rows, mics) with mics None for the single-channel layout (src/models/conv_tasnet.py:127-135)."""
    if input.dim() not in (3, 4):
        raise ValueError("Not support {} dimension input".format(input.dim()))
    assert input.size(1) == 1, "input.size() is expected ({}), but given {}".format(
        "?, 1, ?" if input.dim() == 3 else "?, 1, ?, ?", input.size())
    if input.dim() == 3:
        return input, None
    return input.reshape(input.size(0), input.size(2), input.size(3)), input.size(2)


def decode_sources(decoder, latent, mics, left, right):
    """Every source's masked latent (B, S, N, T') through the synthesis filterbank, the framing zeros cut off again:
    (B, S, T) or (B, S, mics, T) (src/models/conv_tasnet.py:153-159)."""
    B, S = latent.shape[:2]
    x_hat = decoder(latent.flatten(0, 1))
    x_hat = x_hat.reshape(B, S, -1) if mics is None else x_hat.reshape(B, S, mics, -1)
    return x_hat[..., left:x_hat.size(-1) - right]
