"""Layer norms with the reference's interface and state_dict keys (src/modules/norm.py), running on the
HIP statistics / apply kernels (ctn_norm_*)."""
import torch
import torch.nn as nn

from ..functional import LayerNormFn

EPS = 1e-12


class GlobalLayerNorm(nn.Module):
    """Global layer normalization (src/modules/norm.py:11-35).

    The parameters live in an nn.GroupNorm(1, C) container so the state_dict keys stay
    `norm.weight` / `norm.bias`; the container's own forward is never called."""

    def __init__(self, num_features, eps=EPS):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.norm = nn.GroupNorm(1, num_features, eps=eps)

    def forward(self, input):
        """
        Args:
            input (batch_size, C, *)
        Returns:
            output (batch_size, C, *)
        """
        return LayerNormFn.apply(input, self.norm.weight, self.norm.bias, self.eps, False)

    def __repr__(self):
        return '{}({}, eps={})'.format(self.__class__.__name__, self.num_features, self.eps)


class CumulativeLayerNorm1d(nn.Module):
    """Cumulative layer normalization (src/modules/norm.py:42-101); statistics are built on the
    input's device (the reference builds its frame counter on the CPU, norm.py:83)."""

    def __init__(self, num_features, eps=EPS):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.gamma = nn.Parameter(torch.Tensor(1, num_features, 1))
        self.beta = nn.Parameter(torch.Tensor(1, num_features, 1))
        self._reset_parameters()

    def _reset_parameters(self):
        self.gamma.data.fill_(1)
        self.beta.data.fill_(0)

    def forward(self, input):
        """
        Args:
            input (batch_size, C, T) or (batch_size, C, S, chunk_size)
        Returns:
            output: same shape as the input
        """
        if input.dim() not in (3, 4):
            raise ValueError("Only support 3D or 4D input, but given {}D".format(input.dim()))
        return LayerNormFn.apply(input, self.gamma, self.beta, self.eps, True)

    def __repr__(self):
        return '{}({}, eps={})'.format(self.__class__.__name__, self.num_features, self.eps)
