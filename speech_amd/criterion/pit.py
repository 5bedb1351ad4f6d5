"""Permutation invariant training with the reference's interface (src/criterion/pit.py:9-85), one-and-rest PIT
(:87-161) and Sinkhorn PIT (:163-213).

For the SI-SDR criteria the S! candidate losses are never materialised: one fused HIP pass builds the
S x S pairwise SI-SDR matrix, searches the permutations on chip and keeps only the winner's gradient
coefficients (ctn_pit_sisdr_fwd/bwd).  The plain SDR criteria, the 'sum' reduction and Sinkhorn PIT work on the
pairwise matrix of ctn_pair_sdr_fwd/bwd (S^2 passes over the signals' statistics instead of S! or S^2 passes over the
signals).  Any other criterion runs the reference's generic loop.
"""
import itertools

import torch
import torch.nn as nn

from .. import ops
from .sdr import NegSDR, NegSISDR, PairFn, SDR, SISDR


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


def _same_signals(input, target):
    """The kernels walk the target with the estimate's batch, source and sample counts, so both must agree; anything
    else (the reference's generic loop allows a different number of target sources) takes the generic route."""
    return torch.is_tensor(target) and target.shape == input.shape and target.device == input.device


def _fusable(criterion, input, target):
    return (isinstance(criterion, (NegSISDR, SISDR)) and criterion.reduction == 'mean' and input.dim() == 3
            and input.size(1) <= 4 and input.is_cuda and _same_signals(input, target))


def _pair_kind(criterion, input, target):
    """ops.PAIR_* when criterion(input_i, target_j) can be read off the pairwise matrix kernels, else None."""
    if not (input.is_cuda and input.dim() == 3 and input.size(1) <= 4 and _same_signals(input, target)):
        return None
    if type(criterion) in (NegSISDR, SISDR):
        return ops.PAIR_SISDR
    if type(criterion) in (NegSDR, SDR):
        return ops.PAIR_SDR
    return None


def _first_argbest(cand, maximize):
    """Best value per row and the FIRST index that attains it (torch.min / torch.max on the CPU, pit.py:36-39)."""
    best = cand.max(dim=1, keepdim=True)[0] if maximize else cand.min(dim=1, keepdim=True)[0]
    idx = torch.arange(cand.size(1), device=cand.device).expand_as(cand)
    first = torch.where(cand == best, idx, torch.full_like(idx, cand.size(1))).min(dim=1)[0]
    return best.squeeze(1), first


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

    if _fusable(criterion, input, target) and patterns.size(0) == _factorial(input.size(1)):
        loss, perm = _PitSISDRFn.apply(input, target, criterion.eps)
        if maximize:    # SISDR: max over permutations of +mean = -(min of -mean), same winner and tie rule
            loss = -loss
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss, perm

    kind = _pair_kind(criterion, input, target)
    if kind is not None and criterion.reduction in ('mean', 'sum') and input.size(1) == target.size(1) == patterns.size(1):
        S = input.size(1)
        pair = PairFn.apply(input, target, criterion.eps, kind)                     # (B, S, S)
        pat = patterns.to(input.device)                                             # (n_patterns, S)
        cand = pair[:, torch.arange(S, device=input.device).unsqueeze(0), pat]      # (B, n_patterns, S): pair[i][p(i)]
        cand = cand.mean(dim=2) if criterion.reduction == 'mean' else cand.sum(dim=2)
        cand = criterion.sign * cand
        loss, indices = _first_argbest(cand, maximize)
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss, pat[indices]

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


class ORPIT(nn.Module):
    """One-and-rest permutation invariant training (src/criterion/pit.py:87-161): input (batch_size, 2, *) holds one
    separated source and the remainder; every source of the target is tried as the 'one'."""

    def __init__(self, criterion):
        super().__init__()
        self.criterion = criterion
        self.patterns = torch.tensor(list(itertools.permutations(range(2))), dtype=torch.long)

    def forward(self, input, target, batch_mean=True):
        """
        Args:
            input (batch_size, 2, *)
            target (batch_size, n_sources, *) or a PackedSequence of per-sample source lists
        Returns:
            loss (batch_size,) or scalar, indices (batch_size,): which source is the 'one'
        """
        assert input.size(1) == 2, "input.size() is expected (batch_size, 2, *), but given {}".format(input.size())
        criterion = self.criterion
        maximize = hasattr(criterion, "maximize") and criterion.maximize
        if type(target) is torch.Tensor:
            groups = [(slice(0, target.size(0)), target.size(1))]       # one group: the whole batch at once
        else:
            target, lens_unpacked = nn.utils.rnn.pad_packed_sequence(target, batch_first=True)
            groups = [(slice(b, b + 1), int(n)) for b, n in enumerate(lens_unpacked)]
        batch_loss, batch_indices = [], []
        for rows, n_sources in groups:
            _input, _target = input[rows], target[rows, :n_sources]
            input_one, input_rest = torch.unbind(_input, dim=1)
            possible_loss = []
            for idx in range(n_sources):
                target_one = _target[:, idx]
                # the sum of every other source, in index order with a zero in place of the one (pit.py:131-135)
                mask_rest = torch.ones(n_sources, device=_target.device, dtype=_target.dtype)
                mask_rest[idx] = 0.0
                target_rest = torch.sum(mask_rest.view(1, -1, *([1] * (_target.dim() - 2))) * _target, dim=1)
                loss_one = criterion(input_one, target_one, batch_mean=False)
                loss_rest = criterion(input_rest, target_rest, batch_mean=False)
                possible_loss.append(loss_one + loss_rest / (n_sources - 1))
            possible_loss = torch.stack(possible_loss, dim=1)                 # (rows, n_sources)
            loss, indices = _first_argbest(possible_loss, maximize)
            batch_loss.append(loss)
            batch_indices.append(indices)
        batch_loss, batch_indices = torch.cat(batch_loss, dim=0), torch.cat(batch_indices, dim=0)
        if batch_mean:
            batch_loss = batch_loss.mean(dim=0)
        return batch_loss, batch_indices


def sinkpit(criterion, input, target, n_sources=None, coldness=1e+0, iteration=10, batch_mean=True):
    """Sinkhorn PIT (src/criterion/pit.py:163-193): a soft assignment from `iteration` row/column normalisations of
    exp(-coldness * loss matrix).  Returns (loss, permutation matrix (batch_size, n_sources, n_sources))."""
    if n_sources is None:
        n_sources = input.size(1)
    batch_size = input.size(0)
    maximize = hasattr(criterion, "maximize") and criterion.maximize
    kind = _pair_kind(criterion, input, target)
    if kind is not None and input.size(1) == target.size(1) == n_sources:
        # criterion on 2-D inputs has no reduction over sources (sdr.py:49-59): the matrix entries themselves
        possible_loss = criterion.sign * PairFn.apply(input, target, criterion.eps, kind)
    else:
        input_size, target_size = input.size()[2:], target.size()[2:]
        _input = input.unsqueeze(dim=2).expand(-1, -1, n_sources, *input_size).contiguous()
        _target = target.unsqueeze(dim=1).expand(-1, n_sources, -1, *target_size).contiguous()
        _input = _input.view(batch_size * n_sources * n_sources, *input_size)
        _target = _target.view(batch_size * n_sources * n_sources, *target_size)
        possible_loss = criterion(_input, _target, batch_mean=False).view(batch_size, n_sources, n_sources)
    if maximize:
        possible_loss = - possible_loss
    Z = - coldness * possible_loss
    for _ in range(iteration):
        Z = Z - torch.logsumexp(Z, dim=1, keepdim=True)
        Z = Z - torch.logsumexp(Z, dim=2, keepdim=True)
    permutation_matrix = torch.exp(Z)
    loss = torch.sum((possible_loss + Z / coldness) * permutation_matrix, dim=(1, 2))
    if maximize:
        loss = - loss
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, permutation_matrix


class SinkPIT(nn.Module):
    """src/criterion/pit.py:195-213"""

    def __init__(self, criterion, n_sources=None, coldness=1, iteration=10):
        super().__init__()
        self.criterion = criterion
        self.n_sources = n_sources
        self.coldness = coldness
        self.iteration = iteration

    def forward(self, input, target, batch_mean=True):
        loss, permutation_matrix = sinkpit(self.criterion, input, target, n_sources=self.n_sources,
                                           coldness=self.coldness, iteration=self.iteration, batch_mean=batch_mean)
        pattern = torch.argmax(permutation_matrix, dim=2)
        return loss, pattern


class ProbPIT(nn.Module):
    """Placeholder, as in the reference (src/criterion/pit.py:215-225: the forward pass is empty there too)."""

    def __init__(self):
        super().__init__()

    def forward(self, input, target, batch_mean=True):
        pass
