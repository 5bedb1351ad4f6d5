"""Permutation invariant training with the reference's interface (src/criterion/pit.py:9-85).

For the SI-SDR criteria the S! candidate losses are never materialised: one fused HIP pass builds the
S x S pairwise SI-SDR matrix, searches the permutations on chip and keeps only the winner's gradient
coefficients (ctn_pit_sisdr_fwd/bwd).  Any other criterion runs the reference's generic loop.
"""
import itertools

import torch
import torch.nn as nn

from .. import ops
from .sdr import NegSISDR, SISDR


class _PitSISDRFn(torch.autograd.Function):
    """(est, tgt) (B, S, T) -> loss_b (B,) = min_perm -mean_i SI-SDR(est_i, tgt_perm(i)), perm (B, S)."""

    @staticmethod
    def forward(ctx, est, tgt, eps):
        est_c = est.detach().contiguous().float()
        tgt_c = tgt.detach().contiguous().float()
        ops._require_cuda(est_c, tgt_c)
        _, loss_b, perm, coef = ops.pit_sisdr_fwd(est_c, tgt_c, eps)
        ctx.saved = (est_c, tgt_c, perm, coef)
        ctx.mark_non_differentiable(perm)
        return loss_b, perm

    @staticmethod
    def backward(ctx, g, _gperm):
        est_c, tgt_c, perm, coef = ctx.saved
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. the PIT target is not implemented")
        gscale = g.contiguous().float()
        dest = ops.pit_sisdr_bwd(est_c, tgt_c, perm, coef, gscale, 1, 0)
        return dest, None, None


def _fusable(criterion, input):
    return (isinstance(criterion, (NegSISDR, SISDR)) and criterion.reduction == 'mean' and input.dim() == 3
            and input.size(1) <= 4 and input.is_cuda)


def pit(criterion, input, target, n_sources=None, patterns=None, batch_mean=True):
    """
    Args:
        criterion <callable>
        input (batch_size, n_sources, *)
        target (batch_size, n_sources, *)
    Returns:
        loss (batch_size,) or scalar: best loss for each data
        pattern (batch_size, n_sources): permutation indices (applied to the TARGET, pit.py:30)
    """
    if patterns is None:
        if n_sources is None:
            n_sources = input.size(1)
        patterns = torch.tensor(list(itertools.permutations(range(n_sources))), dtype=torch.long)
    maximize = hasattr(criterion, "maximize") and criterion.maximize

    if _fusable(criterion, input) and patterns.size(0) == _factorial(input.size(1)):
        loss, perm = _PitSISDRFn.apply(input, target, criterion.eps)
        if maximize:    # SISDR: max over permutations of +mean = -(min of -mean), same winner and tie rule
            loss = -loss
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss, perm

    possible_loss = []
    for idx in range(len(patterns)):
        possible_loss.append(criterion(input, target[:, patterns[idx].to(target.device)], batch_mean=False))
    possible_loss = torch.stack(possible_loss, dim=1)
    if maximize:
        loss, indices = torch.max(possible_loss, dim=1)
    else:
        loss, indices = torch.min(possible_loss, dim=1)
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, patterns.to(indices.device)[indices]


def _factorial(n):
    r = 1
    for i in range(2, n + 1):
        r *= i
    return r


class PIT(nn.Module):
    def __init__(self, criterion, n_sources):
        """
        Args:
            criterion <callable>: criterion is expected acceptable (input, target, batch_mean) when called.
        """
        super().__init__()
        self.criterion = criterion
        patterns = list(itertools.permutations(range(n_sources)))
        self.patterns = torch.tensor(patterns, dtype=torch.long)   # plain attribute, as in pit.py:55-56

    def forward(self, input, target, batch_mean=True):
        """
        Returns:
            loss (batch_size,) or scalar, pattern (batch_size, n_sources) int64
        """
        return pit(self.criterion, input, target, patterns=self.patterns, batch_mean=batch_mean)


class PIT1d(PIT):
    def __init__(self, criterion, n_sources):
        super().__init__(criterion, n_sources)


class PIT2d(PIT):
    def __init__(self, criterion, n_sources):
        super().__init__(criterion, n_sources)
