"""ctn_finish: a launch that produces a global norm's sums also turns them into mean / rstd or the backward coefficients
(the block that adds the last partial sums does it).  The values must be exactly what the stand-alone kernels compute
from the same sums, for every producer and for launches of one block up to thousands of blocks per batch element."""
import pytest
import torch

from speech_amd.utils import synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def fold_on():
    from speech_amd import ops
    before = ops.FOLD_FINISH
    ops.FOLD_FINISH = ("gemm", "dw")
    yield
    ops.FOLD_FINISH = before


def rnd(shape, seed, scale=1.0):
    return (synth.normal(shape, seed) * scale).float().to(dev())


def _same_as_standalone(acc, stat, B, C, T, eps, coefs):
    from speech_amd import ops
    torch.cuda.synchronize()
    assert (acc.count.cpu() > 0).all()
    if coefs:
        c1, c2 = stat.c1.clone(), stat.c2.clone()
        ops.bwd_coefs(acc, stat, B, C, T, eps)
        torch.cuda.synchronize()
        assert torch.equal(c1, stat.c1) and torch.equal(c2, stat.c2)
    else:
        ref = ops.finalize_stats(acc, B, C, T, eps, dev())
        torch.cuda.synchronize()
        assert torch.equal(stat.mean, ref.mean) and torch.equal(stat.rstd, ref.rstd)
        assert torch.isfinite(stat.rstd).all()


@pytest.mark.parametrize("M,K,T,B", [(512, 128, 3999, 3), (136, 32, 301, 2), (40, 24, 130, 2), (72, 16, 50, 1)])
def test_gemm_stats_epilogue_finishes_its_statistics(M, K, T, B):
    from speech_amd import ops
    A = rnd((K, M), 1)
    X = ops.as_act(rnd((B, K, T), 2))
    Y = ops.new_act(B, M, T, dev())
    alpha = torch.tensor([0.25], device=dev())
    acc = ops.Accum(B, T, False, dev())
    stat, fin = ops.stats_request(acc, B, M, T, 1e-8, dev())
    ops.gemm_nn(M, T, K, B, A, M, X, Y, emode=ops.EPI_STATS, ealpha=alpha, eacc=acc, fin=fin)
    assert int(acc.count.max()) == int(acc.count.min())
    _same_as_standalone(acc, stat, B, M, T, 1e-8, False)


@pytest.mark.parametrize("H,K,T,B", [(512, 256, 3999, 2), (144, 32, 301, 2), (40, 24, 130, 3)])
def test_gemm_normbwd_epilogue_finishes_the_backward_coefficients(H, K, T, B):
    from speech_amd import ops
    A = rnd((K, H), 3)
    X = ops.as_act(rnd((B, K, T), 4))
    y = ops.as_act(rnd((B, H, T), 5) + 0.2)
    alpha = torch.tensor([0.2], device=dev())
    gamma = 1 + 0.2 * rnd((H,), 6)
    acc0 = ops.Accum(B, T, False, dev())
    ops.norm_stats(y, alpha, acc0)
    st = ops.finalize_stats(acc0, B, H, T, 1e-8, dev())
    Y = ops.new_act(B, H, T, dev())
    acc = ops.Accum(B, T, False, dev())
    fin = ops.coefs_request(acc, st, B, H, T)
    ops.gemm_nn(H, T, K, B, A, H, X, Y, emode=ops.EPI_NORMBWD, E=y, ealpha=alpha, estat=st, egamma=gamma, eacc=acc, fin=fin)
    _same_as_standalone(acc, st, B, H, T, 1e-8, True)


@pytest.mark.parametrize("C,T,B,d", [(512, 3999, 2, 1), (512, 3999, 2, 8), (40, 301, 3, 2), (16, 5000, 2, 4)])
def test_depthwise_kernels_finish_their_reductions(C, T, B, d):
    from speech_amd import ops
    P = 3
    x = ops.as_act(rnd((B, C, T), 7) + 0.1)
    w, bias = rnd((C, P), 8), rnd((C,), 9)
    a1, a2 = torch.tensor([0.25], device=dev()), torch.tensor([0.3], device=dev())
    g1, be1, g2 = 1 + 0.2 * rnd((C,), 10), 0.2 * rnd((C,), 11), 1 + 0.2 * rnd((C,), 12)
    acc0 = ops.Accum(B, T, False, dev())
    ops.norm_stats(x, a1, acc0)
    st1 = ops.finalize_stats(acc0, B, C, T, 1e-8, dev())
    y = ops.new_act(B, C, T, dev())
    acc = ops.Accum(B, T, False, dev())
    st2, fin = ops.stats_request(acc, B, C, T, 1e-8, dev(), producer="dw")
    ops.dw_fwd(x, w, bias, y, d, d, xflags=ops.XF_PRELU | ops.XF_NORM, alpha_in=a1, stat=st1, gamma=g1, beta=be1,
               alpha_out=a2, acc=acc, fin=fin)
    _same_as_standalone(acc, st2, B, C, T, 1e-8, False)
    # backward: the norm-1 coefficients from the depthwise backward kernel (stored and recomputed convolution output)
    dn2 = ops.as_act(rnd((B, C, T), 13))
    st2.c1, st2.c2 = 0.01 * rnd((B,), 14), 0.01 * rnd((B,), 15)
    for recompute in (False, True):
        dn1 = ops.new_act(B, C, T, dev())
        dw, db = torch.zeros(C, P, device=dev()), torch.zeros(C, device=dev())
        dg1, db1, dg2, db2 = (torch.zeros(C, device=dev()) for _ in range(4))
        da2 = torch.zeros(ops._lib.STAT_SLOTS, device=dev(), dtype=torch.float64)
        bacc = ops.Accum(B, T, False, dev())
        finb = ops.coefs_request(bacc, st1, B, C, T, producer="dw")
        ops.dw_bwd(dn2, x, w, dw, db, dn1, d, d, xflags=ops.XF_PRELU | ops.XF_NORM, alpha1=a1, stat1=st1, gamma1=g1,
                   beta1=be1, dgamma1=dg1, dbeta1=db1, iacc=bacc, y2=y, alpha2=a2, stat2=st2, gamma2=g2, dgamma2=dg2,
                   dbeta2=db2, dalpha2=da2, bias=bias, recompute_y2=recompute, fin=finb)
        _same_as_standalone(bacc, st1, B, C, T, 1e-8, True)


def test_whole_network_gradients_do_not_depend_on_where_the_reductions_finish():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import convtasnet_oracle as O
    from bench import build_model
    from conftest import assert_close
    from speech_amd import ops
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.trainer import FlatTrainer
    cfg = dict(O.CONFIG_PAPER, sep_num_blocks=2, sep_num_layers=3)
    mix, src = synth.make_batch(3, 2, 8000, 5)
    mix, src = mix.to(dev()), src.to(dev())
    outs = []
    for fold in (("gemm", "dw"), ()):
        ops.FOLD_FINISH = fold
        try:
            tr = FlatTrainer(build_model(cfg, dev()), PIT1d(NegSISDR(), 2))
            loss, _ = tr.forward_backward(mix, src)
            torch.cuda.synchronize()
            outs.append((loss.item(), tr.flat_g.clone()))
        finally:
            ops.FOLD_FINISH = ("gemm", "dw")  # (the fixture restores the module's own default)
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * max(1.0, abs(outs[1][0]))
    assert_close(outs[0][1], outs[1][1], 2e-5, "flat gradient")
