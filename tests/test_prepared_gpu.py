"""Prepared weights (ctn_weight_prep + the A_prep operand of ctn_gemm_nn): the planes hold an exact three-piece split in
the documented layout, and a GEMM fed with them returns bit for bit what the same call returns when the kernel splits
the fp32 weights itself -- in every operand-transform / epilogue combination the training step uses."""
import pytest
import torch

from conftest import assert_close
from speech_amd.utils import synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


def dev():
    return torch.device("cuda:0")


@pytest.fixture(params=[1, 2, 3], ids=["wave_split", "block_split", "wave_split8"], autouse=True)
def prep_kernel(request):
    """Every test runs once per kernel that consumes the prepared operand (CTN_PREP_WAVE_SPLIT, CTN_PREP_BLOCK_SPLIT,
    CTN_PREP_WAVE_SPLIT8; the last one has its own tile width and ignores the tests' width parameter)."""
    from speech_amd import ops
    if ops.get_gemm_split() != 3:
        pytest.skip("a prepared operand exists in the three-piece arithmetic only (CTN_GEMM_SPLIT selects another)")
    before = ops._prep_kernel
    ops._prep_kernel = request.param
    yield request.param
    ops._prep_kernel = before


def rnd(shape, seed, scale=1.0):
    return synth.normal(shape, seed).double() * scale


def f(t):
    return t.float().to(dev())


def act(t):
    from speech_amd import ops
    return ops.as_act(t.to(dev()).float())


def prelu(x, a):
    return torch.where(x > 0, x, a * x)


def decode_planes(planes, K, M):
    """(3, K, M) float64 pieces from a prepared matrix."""
    Kp, Mp = (K + 15) // 16 * 16, (M + 127) // 128 * 128
    p = planes.cpu().view(3, Kp // 8, Mp, 8).double()              # [piece][k-group][m][k in group]
    return p.permute(0, 1, 3, 2).reshape(3, Kp, Mp)


@pytest.mark.parametrize("K,M", [(128, 512), (144, 200), (24, 72)])
def test_weight_prep_is_an_exact_split_in_fragment_layout(K, M):
    from speech_amd import ops
    W = rnd((M, K), 1).float()                                       # Conv1d layout (out, in)
    W[0, 0], W[1, 1], W[2, 2] = 0.0, 1e-30, -3.0e20                  # zero, a tiny and a huge value
    Wd = W.to(dev())
    fwd = ops.prepared(K, M, [(Wd, True, 0, 0)], dev())              # A[k = in][m = out]
    bwd = ops.prepared(M - M % 8, K, [(Wd[:M - M % 8], False, 0, 0)], dev())   # A[k = out][m = in]
    torch.cuda.synchronize()
    pf = decode_planes(fwd, K, M)
    assert torch.equal(pf.sum(0)[:K, :M], W.t().double())            # hi + mid + lo == x exactly
    assert pf[:, K:, :].abs().sum() == 0 and pf[:, :, M:].abs().sum() == 0
    # every piece is a bf16 value holding at most eight significant bits of the fp32 number, pieces ordered by size
    assert (pf[0].abs() >= pf[1].abs()).all() and (pf[1].abs() >= pf[2].abs()).all()
    Mk = M - M % 8
    pb = decode_planes(bwd, Mk, K)
    assert torch.equal(pb.sum(0)[:Mk, :K], W[:Mk].double())


def test_weight_prep_places_two_sources_side_by_side_and_stacked():
    from speech_amd import ops
    H, Bc, Sc = 144, 64, 80
    Wo, Ws = rnd((Bc, H), 2).float().to(dev()), rnd((Sc, H), 3).float().to(dev())
    heads_f = ops.prepared(H, Bc + Sc, [(Wo, True, 0, 0), (Ws, True, 0, Bc)], dev())
    heads_d = ops.prepared(Bc + Sc, H, [(Wo, False, 0, 0), (Ws, False, Bc, 0)], dev())
    torch.cuda.synchronize()
    A = decode_planes(heads_f, H, Bc + Sc).sum(0)[:H, :Bc + Sc]
    assert torch.equal(A, torch.cat([Wo, Ws], 0).t().double().cpu())
    A = decode_planes(heads_d, Bc + Sc, H).sum(0)[:Bc + Sc, :H]
    assert torch.equal(A, torch.cat([Wo, Ws], 0).double().cpu())


def test_weight_prep_rejects_bad_items():
    from speech_amd import ops, _lib
    W = torch.zeros(32, 32, device=dev())
    dst = torch.zeros(ops.weight_planes_numel(32, 32), device=dev(), dtype=torch.bfloat16)
    with pytest.raises(_lib.CtnError):
        ops.weight_prep(ops.wprep_items([(W, False, 4, 0, 64, 32, torch.zeros(ops.weight_planes_numel(64, 32), device=dev(),
                                                                               dtype=torch.bfloat16))]))   # k_off % 8
    with pytest.raises(_lib.CtnError):
        ops.weight_prep(ops.wprep_items([(W, False, 8, 0, 32, 32, dst)]))                                   # does not fit


def _both(call):
    """Runs call(A_prep) with the prepared operand and without; returns the two result lists."""
    return call(True), call(False)


@pytest.mark.parametrize("width", [64, 128])
@pytest.mark.parametrize("M,K,T,B", [(512, 128, 3999, 1), (136, 32, 301, 2), (72, 16, 130, 3), (256, 512, 1000, 1)])
def test_gemm_plain_and_stats_epilogue(M, K, T, B, width):
    from speech_amd import ops
    W, bias, X = rnd((M, K), 10), rnd((M,), 11), rnd((B, K, T), 12)
    alpha = torch.tensor([0.25], dtype=torch.float64)
    ref = torch.einsum("mk,bkt->bmt", W, X) + bias.view(1, M, 1)
    Wd = f(W)
    Wt = torch.empty(K, M, device=dev())
    ops.transpose(Wd, Wt, M)
    Wp = ops.prepared(K, M, [(Wd, True, 0, 0)], dev())
    Xa = act(X)

    def call(prep):
        Y = ops.new_act(B, M, T, dev())
        acc = ops.Accum(B, T, False, dev())
        ops.gemm_nn(M, T, K, B, None if prep else Wt, M, Xa, Y, bias=f(bias), emode=ops.EPI_STATS, ealpha=f(alpha),
                    eacc=acc, tile_width=width, A_prep=Wp if prep else None)
        torch.cuda.synchronize()
        return Y[:, :, :T].clone(), acc.sums.sum(1)

    (Yp, Sp), (Yk, Sk) = _both(call)
    assert torch.equal(Yp, Yk)
    assert_close(Yp, ref, TOL, "Y")
    assert_close(Sp, Sk, 1e-6, "sums")


@pytest.mark.parametrize("width", [64, 128])
@pytest.mark.parametrize("cln", [False, True])
def test_gemm_norm_on_load_heads(cln, width):
    from speech_amd import ops
    from test_kernels_gpu import gln_stats, cln_stats, make_stat
    H, Bc, Sc, T, B = 48, 72, 64, 301, 2
    y2 = rnd((B, H, T), 20) + 0.3
    alpha = torch.tensor([0.2], dtype=torch.float64)
    gamma, beta = 1 + 0.2 * rnd((H,), 21), 0.2 * rnd((H,), 22)
    u = prelu(y2, alpha)
    mean, rstd = (cln_stats if cln else gln_stats)(u)
    n = (u - mean) * rstd * gamma.view(1, H, 1) + beta.view(1, H, 1)
    Wo, Ws, bo, bs = rnd((Bc, H), 23), rnd((Sc, H), 24), rnd((Bc,), 25), rnd((Sc,), 26)
    x, skip0 = rnd((B, Bc, T), 27), rnd((B, Sc, T), 28)
    ref_out = torch.einsum("mk,bkt->bmt", Wo, n) + bo.view(1, -1, 1) + x
    ref_skip = torch.einsum("mk,bkt->bmt", Ws, n) + bs.view(1, -1, 1) + skip0
    Wct = torch.empty(H, Bc + Sc, device=dev())
    ops.transpose(f(Wo), Wct, Bc + Sc, 0)
    ops.transpose(f(Ws), Wct, Bc + Sc, Bc)
    Wp = ops.prepared(H, Bc + Sc, [(f(Wo), True, 0, 0), (f(Ws), True, 0, Bc)], dev())
    st = make_stat(mean, rstd, cln)

    def call(prep):
        out = ops.new_act(B, Bc, T, dev())
        skip = act(skip0.clone())
        ops.gemm_nn(Bc + Sc, T, H, B, None if prep else Wct, Bc + Sc, act(y2), out, xflags=ops.XF_PRELU | ops.XF_NORM,
                    alpha=f(alpha), stat=st, gamma=f(gamma), beta=f(beta), bias=f(bo), bias2=f(bs), emode=ops.EPI_HEADS,
                    Mo=Bc, R=act(x), Y2=skip, accumulate=True, tile_width=width, A_prep=Wp if prep else None)
        torch.cuda.synchronize()
        return out[:, :, :T].clone(), skip[:, :, :T].clone()

    (op_, sp_), (ok_, sk_) = _both(call)
    assert torch.equal(op_, ok_) and torch.equal(sp_, sk_)
    assert_close(op_, ref_out, TOL, "heads out")
    assert_close(sp_, ref_skip, TOL, "heads skip")


@pytest.mark.parametrize("width", [64, 128])
@pytest.mark.parametrize("cln", [False, True])
def test_gemm_split_k_normbwd_epilogue_then_normbwd_on_load(cln, width):
    from speech_amd import ops
    from test_kernels_gpu import gln_stats, cln_stats, make_stat
    H, K1, K2, T, B = 144, 64, 80, 777, 2            # more skip than residual channels, ragged length
    Wo, Ws = rnd((K1, H), 30), rnd((K2, H), 31)      # (out, in) of the two heads: k-major for their input gradient
    Xa, Xb = rnd((B, K1, T), 32), rnd((B, K2, T), 33)
    y = rnd((B, H, T), 34) + 0.2
    alpha = torch.tensor([0.2], dtype=torch.float64)
    gamma = 1 + 0.2 * rnd((H,), 35)
    u = prelu(y, alpha)
    mean, rstd = (cln_stats if cln else gln_stats)(u)
    dn = torch.einsum("km,bkt->bmt", Wo, Xa) + torch.einsum("km,bkt->bmt", Ws, Xb)
    st = make_stat(mean, rstd, cln)
    Wp = ops.prepared(K1 + K2, H, [(f(Wo), False, 0, 0), (f(Ws), False, K1, 0)], dev())
    M2 = 72
    W2 = rnd((H, M2), 36)                             # (out = H, in = M2): k-major for the input gradient
    R = rnd((B, M2, T), 37)
    W2p = ops.prepared(H, M2, [(f(W2), False, 0, 0)], dev())

    def call(prep):
        Y = ops.new_act(B, H, T, dev())
        acc = ops.Accum(B, T, cln, dev())
        ops.gemm_nn(H, T, K1 + K2, B, None if prep else f(Wo), H, act(Xa), Y, A2=None if prep else f(Ws), lda2=H, K1=K1,
                    Xb=act(Xb), emode=ops.EPI_NORMBWD, E=act(y), ealpha=f(alpha), estat=st, egamma=f(gamma), eacc=acc,
                    tile_width=width, A_prep=Wp if prep else None)
        ops.bwd_coefs(acc, st, B, H, T, 1e-12)
        Y2 = ops.new_act(B, M2, T, dev())
        ops.gemm_nn(M2, T, H, B, None if prep else f(W2), M2, Y, Y2, xflags=ops.XF_NORMBWD, X2=act(y), alpha=f(alpha),
                    stat=st, gamma=f(gamma), emode=ops.EPI_ADD, R=act(R), tile_width=width, A_prep=W2p if prep else None)
        torch.cuda.synchronize()
        return Y[:, :, :T].clone(), Y2[:, :, :T].clone()

    (Yp, Y2p), (Yk, Y2k) = _both(call)
    assert torch.equal(Yp, Yk)
    assert_close(Yp, dn, TOL, "dn")
    # the second GEMM's statistics come from sums accumulated atomically in the first: equal up to their summation order
    assert_close(Y2p, Y2k, 1e-5, "normbwd on load, prepared vs in-kernel split")
    yv = y.clone().requires_grad_(True)
    uu = prelu(yv, alpha)
    m2, r2 = (cln_stats if cln else gln_stats)(uu)
    ((uu - m2) * r2 * gamma.view(1, H, 1)).backward(dn)
    assert_close(Y2p, torch.einsum("km,bkt->bmt", W2, yv.grad) + R, 1e-4, "normbwd on load")


def test_gemm_sigmoid_and_prelu_bwd_epilogues():
    from speech_amd import ops
    M, K, T, B = 132, 32, 130, 2
    W, bias, X = rnd((M, K), 40), rnd((M,), 41), rnd((B, K, T), 42)
    alpha = torch.tensor([0.3], dtype=torch.float64)
    Wp = ops.prepared(K, M, [(f(W), True, 0, 0)], dev())
    Y = ops.new_act(B, M, T, dev())
    ops.gemm_nn(M, T, K, B, None, M, act(X), Y, xflags=ops.XF_PRELU, alpha=f(alpha), bias=f(bias), emode=ops.EPI_SIGMOID,
                A_prep=Wp)
    torch.cuda.synchronize()
    assert_close(Y, torch.sigmoid(torch.einsum("mk,bkt->bmt", W, prelu(X, alpha)) + bias.view(1, M, 1)), TOL, "sigmoid")
    E = rnd((B, M, T), 43)
    acc = torch.einsum("mk,bkt->bmt", W, X)
    da = torch.zeros(ops._lib.STAT_SLOTS, device=dev(), dtype=torch.float64)
    ops.gemm_nn(M, T, K, B, None, M, act(X), Y, emode=ops.EPI_PRELUBWD, E=act(E), ealpha=f(alpha), esums=da, A_prep=Wp)
    torch.cuda.synchronize()
    assert_close(Y, torch.where(E > 0, acc, alpha * acc), TOL, "prelu bwd")
    assert_close(da.sum().reshape(1), (acc * torch.where(E > 0, torch.zeros_like(E), E)).sum().reshape(1), 1e-4, "dalpha")


def test_gemm_refuses_a_prepared_operand_it_cannot_take():
    from speech_amd import ops, _lib
    M, K, T, B = 132, 24, 130, 1                      # K % 16 != 0
    Wp = torch.zeros(ops.weight_planes_numel(K, M), device=dev(), dtype=torch.bfloat16)
    Y = ops.new_act(B, M, T, dev())
    with pytest.raises(_lib.CtnError):
        ops.gemm_nn(M, T, K, B, None, M, act(rnd((B, K, T), 50)), Y, A_prep=Wp)


def test_trainer_gradients_agree_with_and_without_prepared_weights():
    """One forward/backward pass of a small network through FlatTrainer with its prepared buffer (refreshed per pass)
    and with the in-kernel split: loss and the flat gradient agree to fp32 summation-order noise (the atomics)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import convtasnet_oracle as O
    from bench import build_model
    from speech_amd import functional
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.trainer import FlatTrainer
    cfg = dict(O.CONFIG_PAPER, sep_num_blocks=1, sep_num_layers=3)
    mix, src = synth.make_batch(2, 2, 8000, 5)
    mix, src = mix.to(dev()), src.to(dev())
    outs = []
    before = functional.USE_PREPARED_WEIGHTS
    for use in (True, False):
        functional.USE_PREPARED_WEIGHTS = use
        try:
            tr = FlatTrainer(build_model(cfg, dev()), PIT1d(NegSISDR(), 2))
            tr.forward_backward(mix, src)
            tr.flat_p.mul_(1.02)                            # the prepared buffer must follow the weights (an optimizer
            loss, _ = tr.forward_backward(mix, src)         # step itself would amplify summation noise: Adam normalises)
            torch.cuda.synchronize()
            outs.append((loss.item(), tr.flat_g.clone()))
        finally:
            functional.USE_PREPARED_WEIGHTS = before
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * max(1.0, abs(outs[1][0]))
    assert_close(outs[0][1], outs[1][1], 2e-5, "flat gradient")
