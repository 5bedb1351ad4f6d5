"""ctn_pit_sisdr_fwd_noalloc (the entry point that allocates nothing) against ctn_pit_sisdr_fwd, through the C ABI.

Two calls in a row on the same GPU, the second with a smaller input: the built-in work area keeps the first call's sums,
so the second call is right only if the entry point clears what it uses.  Permutations are held bit-exact; losses,
pairwise values and gradient coefficients to 1e-6 relative (both paths add fp64 partial sums with atomics, whose order
differs from launch to launch).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(name, est, tgt, fixed_perm=False, eps=1e-12):
    from speech_amd import _lib
    lib = _lib.load()
    B, S, T = est.shape
    dev = est.device
    pair = torch.empty((B, S, S), device=dev, dtype=torch.float32)
    loss_b = torch.empty((B,), device=dev, dtype=torch.float32)
    perm = torch.empty((B, S), device=dev, dtype=torch.int64)
    coef = torch.empty((B, S, 2), device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    head = (est.data_ptr(), tgt.data_ptr(), B, S, T, est.stride(1), est.stride(0), eps, 1 if fixed_perm else 0)
    tail = (pair.data_ptr(), loss_b.data_ptr(), perm.data_ptr(), coef.data_ptr(), stream)
    if name == "ctn_pit_sisdr_fwd":
        work = torch.zeros((B, 2 * S * S + S), device=dev, dtype=torch.float64)
        rc = lib.ctn_pit_sisdr_fwd(*head, work.data_ptr(), *tail)
    else:
        rc = lib.ctn_pit_sisdr_fwd_noalloc(*head, *tail)
    return rc, (pair, loss_b, perm, coef)


def _signals(B, S, T, seed):
    g = torch.Generator().manual_seed(seed)
    tgt = 0.1 * torch.randn(B, S, T, generator=g)
    est = tgt[:, torch.randperm(S, generator=g)] + 0.03 * torch.randn(B, S, T, generator=g)
    return est.cuda(), tgt.cuda()


def _same(a, b):
    for x, y, exact in zip(a, b, (False, False, True, False)):
        if exact:
            assert torch.equal(x, y)
        else:
            scale = y.abs().max().item()
            assert (x - y).abs().max().item() <= 1e-6 * scale, ((x - y).abs().max().item(), scale)


@pytest.mark.parametrize("S", [2, 3])
def test_noalloc_matches_workspace_entry_point_twice(S):
    torch.cuda.init()
    for B, T, seed in ((8, 32000, 1), (3, 9001, 2)):      # second call smaller: stale sums of the first must be cleared
        est, tgt = _signals(B, S, T, seed)
        rc_ref, ref = _call("ctn_pit_sisdr_fwd", est, tgt)
        rc_new, new = _call("ctn_pit_sisdr_fwd_noalloc", est, tgt)
        torch.cuda.synchronize()
        assert rc_ref == 0 and rc_new == 0
        _same(new, ref)


def test_noalloc_rejects_oversized_batch():
    from speech_amd import _lib
    lib = _lib.load()
    S = 4
    B = _lib.PIT_NOALLOC_MAX_WORK // (2 * S * S + S) + 1
    est = torch.zeros((B, S, 8), device="cuda")
    rc, _ = _call("ctn_pit_sisdr_fwd_noalloc", est, est)
    assert rc == -1
    assert b"exceeds" in lib.ctn_last_error()
    # and the largest batch that fits is accepted
    est = torch.randn((B - 1, S, 8), device="cuda")
    rc_new, new = _call("ctn_pit_sisdr_fwd_noalloc", est, est.flip(1))
    rc_ref, ref = _call("ctn_pit_sisdr_fwd", est, est.flip(1))
    torch.cuda.synchronize()
    assert rc_new == 0 and rc_ref == 0
    _same(new, ref)
