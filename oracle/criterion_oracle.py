"""CPU restatement of the reference's criterion variants beside the hot path: plain SDR, one-and-rest PIT and
Sinkhorn PIT.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and never by the speech_amd package); every function
cites the reference lines it follows.  Pinned by tests/golden/criterion.npz, which tools/gen_golden_criterion.py wrote
by running the reference's own src/criterion modules.  Works in any floating dtype (float64 for tight comparisons).
"""
import torch

EPS = 1e-12


def sdr(est, tgt, eps=EPS):
    """sdr: src/criterion/sdr.py:19-20 -- target energy over the energy of the plain difference."""
    err = tgt - est
    return 10 * torch.log10(((tgt * tgt).sum(dim=-1) + eps) / ((err * err).sum(dim=-1) + eps))


def sisdr(est, tgt, eps=EPS):
    """sisdr: src/criterion/sdr.py:135-137."""
    alpha = (est * tgt).sum(dim=-1, keepdim=True) / ((tgt * tgt).sum(dim=-1, keepdim=True) + eps)
    proj = alpha * tgt
    return 10 * torch.log10(((proj * proj).sum(dim=-1) + eps) / (((proj - est) ** 2).sum(dim=-1) + eps))


class Criterion:
    """SDR / NegSDR / SISDR / NegSISDR .forward: src/criterion/sdr.py:35-64, :81-110, :152-181, :198-227 -- the value,
    negated for the Neg* classes, reduced over the source (and microphone) axes for 3-D (4-D) inputs, then over the
    batch.  `maximize` as the properties at :66-68, :112-114, :183-185, :229-231."""

    def __init__(self, fn, negate, reduction="mean", eps=EPS):
        self.fn, self.negate, self.reduction, self.eps = fn, negate, reduction, eps
        self.maximize = not negate

    def __call__(self, est, tgt, batch_mean=True):
        loss = self.fn(est, tgt, self.eps)
        if self.negate:
            loss = -loss
        if self.reduction and est.dim() >= 3:
            dims = tuple(range(1, est.dim() - 1))
            loss = loss.mean(dim=dims) if self.reduction == "mean" else loss.sum(dim=dims)
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss


def orpit(criterion, est, tgt_list, batch_mean=True):
    """ORPIT.forward: src/criterion/pit.py:98-161.  est (B, 2, T); tgt_list: per sample a (n_sources_b, T) tensor
    (what pad_packed_sequence + lens give, :116, :124-125).  For every candidate 'one' the loss is
    criterion(one) + criterion(rest) / (n_sources - 1) (:137-139); best candidate per sample with first-index ties
    (:146-149)."""
    losses, picks = [], []
    for b, tgt in enumerate(tgt_list):
        n = tgt.shape[0]
        cand = []
        for k in range(n):
            one = tgt[k:k + 1]
            rest = sum(tgt[j] for j in range(n) if j != k).unsqueeze(0)
            cand.append(criterion(est[b:b + 1, 0], one, batch_mean=False)
                        + criterion(est[b:b + 1, 1], rest, batch_mean=False) / (n - 1))
        cand = torch.cat(cand, dim=0)
        best = int(torch.argmax(cand)) if criterion.maximize else int(torch.argmin(cand))
        # first index among exact ties
        for k in range(n):
            if cand[k] == cand[best]:
                best = k
                break
        losses.append(cand[best])
        picks.append(best)
    loss = torch.stack(losses)
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, torch.tensor(picks, dtype=torch.long)


def sinkpit(criterion, est, tgt, coldness=1.0, iteration=10, batch_mean=True):
    """sinkpit: src/criterion/pit.py:163-193.  The matrix of criterion(est_i, tgt_j) on 2-D inputs (:170-173), sign
    flipped for a criterion to maximise (:175-176); Z = -coldness * matrix normalised `iteration` times over dim 1 then
    dim 2 in the log domain (:178-182); loss = sum((matrix + Z / coldness) * exp(Z)) (:184-185)."""
    B, S = est.shape[:2]
    rows = []
    for i in range(S):
        rows.append(torch.stack([criterion(est[:, i], tgt[:, j], batch_mean=False) for j in range(S)], dim=1))
    M = torch.stack(rows, dim=1)                       # (B, S, S)
    if criterion.maximize:
        M = -M
    Z = -coldness * M
    for _ in range(iteration):
        Z = Z - torch.logsumexp(Z, dim=1, keepdim=True)
        Z = Z - torch.logsumexp(Z, dim=2, keepdim=True)
    P = torch.exp(Z)
    loss = ((M + Z / coldness) * P).sum(dim=(1, 2))
    if criterion.maximize:
        loss = -loss
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, P
