"""SDR and SI-SDR criteria with the reference's interface (src/criterion/sdr.py:6-231), computed by the fused HIP
kernels: SI-SDR by ctn_pit_sisdr_fwd/bwd with the identity assignment, the plain SDR and every pairwise matrix by
ctn_pair_sdr_fwd/bwd."""
import torch
import torch.nn as nn

from .. import ops

EPS = 1e-12


class _SISDRFn(torch.autograd.Function):
    """est, tgt (B, S, T) -> per-pair SI-SDR in dB, (B, S) for the identity assignment."""

    @staticmethod
    def forward(ctx, est, tgt, eps):
        est_c = est.detach().contiguous().float()
        tgt_c = tgt.detach().contiguous().float()
        ops._require_cuda(est_c, tgt_c)
        pair, _, perm, coef = ops.pit_sisdr_fwd(est_c, tgt_c, eps, fixed_perm=True)
        ctx.saved = (est_c, tgt_c, perm, coef)
        return torch.diagonal(pair, dim1=1, dim2=2).contiguous()

    @staticmethod
    def backward(ctx, g):
        est_c, tgt_c, perm, coef = ctx.saved
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. the SI-SDR target is not implemented")
        S = est_c.shape[1]
        # coef holds d(-mean_i sisdr_i)/d est_i = -(1/S) d sisdr_i / d est_i
        gscale = (g.contiguous().float() * (-float(S))).contiguous()
        dest = ops.pit_sisdr_bwd(est_c, tgt_c, perm, coef, gscale, S, 1)
        return dest, None, None


class PairFn(torch.autograd.Function):
    """est, tgt (B, S, T), S <= 4 -> (B, S, S): criterion(est_i, tgt_j) in dB for every pair, kind = ops.PAIR_SISDR or
    ops.PAIR_SDR.  Differentiable in est: the backward pass weights every pair's derivative with its upstream gradient."""

    @staticmethod
    def forward(ctx, est, tgt, eps, kind):
        est_c = est.detach().contiguous().float()
        tgt_c = tgt.detach().contiguous().float()
        ops._require_cuda(est_c, tgt_c)
        pair, paircoef = ops.pair_sdr_fwd(est_c, tgt_c, eps, kind)
        ctx.saved = (est_c, tgt_c, paircoef)
        return pair

    @staticmethod
    def backward(ctx, g):
        est_c, tgt_c, paircoef = ctx.saved
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. the target is not implemented")
        g = g.float()
        A = (g * paircoef[..., 0]).contiguous()
        Cc = (g * paircoef[..., 1]).sum(dim=2).contiguous()
        return ops.pair_sdr_bwd(est_c, tgt_c, A, Cc), None, None, None


def _as_bst(x):
    """(B, T) / (B, S, T) / (B, S, M, T) -> (B, S', T) and the shape of the per-source result."""
    n = x.dim()
    assert n in [2, 3, 4], "Only 2D or 3D or 4D tensor is acceptable, but given {}D tensor.".format(n)
    if n == 2:
        return x.unsqueeze(1), x.shape[:1]
    if n == 3:
        return x, x.shape[:2]
    return x.reshape(x.shape[0], -1, x.shape[-1]), x.shape[:3]


def sisdr(input, target, eps=EPS):
    """Scale-invariant SDR in dB (src/criterion/sdr.py:122-139).
    Args:
        input, target: (batch_size, T), (batch_size, n_sources, T) or (batch_size, n_sources, n_mics, T)
    Returns:
        (batch_size,), (batch_size, n_sources) or (batch_size, n_sources, n_mics)
    """
    est, out_shape = _as_bst(input)
    tgt, _ = _as_bst(target)
    S = est.shape[1]
    if S > 4:   # the kernel handles up to 4 rows per call
        chunks = [_SISDRFn.apply(est[:, i:i + 4], tgt[:, i:i + 4], eps) for i in range(0, S, 4)]
        loss = torch.cat(chunks, dim=1)
    else:
        loss = _SISDRFn.apply(est, tgt, eps)
    return loss.reshape(out_shape)


class _SISDRBase(nn.Module):
    sign = 1.0

    def __init__(self, reduction='mean', eps=EPS):
        super().__init__()
        self.reduction = reduction
        if reduction not in ['mean', 'sum', None]:
            raise ValueError("Invalid reduction type")
        self.eps = eps

    def forward(self, input, target, batch_mean=True):
        n_dims = input.dim()
        assert n_dims in [2, 3, 4], "Only 2D or 3D or 4D tensor is acceptable, but given {}D tensor.".format(n_dims)
        loss = self.sign * sisdr(input, target, eps=self.eps)
        if self.reduction:
            if n_dims == 3:
                loss = loss.mean(dim=1) if self.reduction == 'mean' else loss.sum(dim=1)
            elif n_dims == 4:
                loss = loss.mean(dim=(1, 2)) if self.reduction == 'mean' else loss.sum(dim=(1, 2))
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss


class SISDR(_SISDRBase):
    """src/criterion/sdr.py:141-185"""
    sign = 1.0

    @property
    def maximize(self):
        return True


class NegSISDR(_SISDRBase):
    """src/criterion/sdr.py:187-231"""
    sign = -1.0

    @property
    def maximize(self):
        return False


def sdr(input, target, eps=EPS):
    """Source-to-distortion ratio in dB, 10 log10((|t|^2 + eps) / (|t - e|^2 + eps)) (src/criterion/sdr.py:6-22).
    Args:
        input, target: (batch_size, T), (batch_size, n_sources, T) or (batch_size, n_sources, n_mics, T)
    Returns:
        (batch_size,), (batch_size, n_sources) or (batch_size, n_sources, n_mics)
    """
    est, out_shape = _as_bst(input)
    tgt, _ = _as_bst(target)
    S = est.shape[1]
    chunks = [torch.diagonal(PairFn.apply(est[:, i:i + 4], tgt[:, i:i + 4], eps, ops.PAIR_SDR), dim1=1, dim2=2)
              for i in range(0, S, 4)]        # the kernel handles up to 4 rows per call
    loss = chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=1)
    return loss.reshape(out_shape)


class _SDRBase(_SISDRBase):
    def forward(self, input, target, batch_mean=True):
        n_dims = input.dim()
        assert n_dims in [2, 3, 4], "Only 2D or 3D or 4D tensor is acceptable, but given {}D tensor.".format(n_dims)
        loss = self.sign * sdr(input, target, eps=self.eps)
        if self.reduction:
            if n_dims == 3:
                loss = loss.mean(dim=1) if self.reduction == 'mean' else loss.sum(dim=1)
            elif n_dims == 4:
                loss = loss.mean(dim=(1, 2)) if self.reduction == 'mean' else loss.sum(dim=(1, 2))
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss


class SDR(_SDRBase):
    """src/criterion/sdr.py:24-68"""
    sign = 1.0

    @property
    def maximize(self):
        return True


class NegSDR(_SDRBase):
    """src/criterion/sdr.py:70-114"""
    sign = -1.0

    @property
    def maximize(self):
        return False
