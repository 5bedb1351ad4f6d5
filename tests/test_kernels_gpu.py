"""Kernel-level parity: each C-ABI entry point (through speech_amd.ops) against the same arithmetic in
torch float64 on the CPU.  Tolerances are relative to the largest reference magnitude (fp32 kernels)."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from speech_amd.utils import synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


def dev():
    return torch.device("cuda:0")


# "generic": dimensions the register-staged GEMM serves; "fast64"/"fast128": dimensions that qualify for the
# LDS-DMA kernel (M > 64, M % 4 == 0, K % 16 == 0), with its output-tile width forced to 64 / 128 columns.
PATHS = ["generic", "fast64", "fast128"]


@pytest.fixture
def gemm_path(request):
    from speech_amd import ops
    path = request.param
    ops.set_gemm_tile({"generic": 0, "fast64": 64, "fast128": 128}[path])
    yield path
    ops.set_gemm_tile(0)


def act(t):
    from speech_amd import ops
    return ops.as_act(t.to(dev()).float())


def rnd(shape, seed, scale=1.0):
    return synth.normal(shape, seed).double() * scale


def prelu(x, a):
    return torch.where(x > 0, x, a * x)


def gln_stats(u, eps=1e-12):
    mean = u.mean(dim=(1, 2), keepdim=True)
    var = (u ** 2).mean(dim=(1, 2), keepdim=True) - mean ** 2
    return mean, 1.0 / torch.sqrt(var + eps)


def cln_stats(u, eps=1e-12):
    B, C, T = u.shape
    cnt = torch.arange(1, T + 1, dtype=u.dtype) * C
    mean = torch.cumsum(u.sum(1), 1) / cnt
    var = torch.cumsum((u ** 2).sum(1), 1) / cnt - mean ** 2
    return mean.unsqueeze(1), 1.0 / (torch.sqrt(var.unsqueeze(1)) + eps)


def make_stat(mean, rstd, cln):
    from speech_amd import ops
    B = mean.shape[0]
    if cln:
        T = mean.shape[-1]
        ld = ops.round4(T)
        m = torch.zeros(B, ld)
        r = torch.ones(B, ld)
        m[:, :T] = mean.reshape(B, T)
        r[:, :T] = rstd.reshape(B, T)
        return ops.Stat(m.to(dev()).float(), r.to(dev()).float(), ld, 1, True)
    return ops.Stat(mean.reshape(B).to(dev()).float(), rstd.reshape(B).to(dev()).float(), 1, 0, False)


def test_transpose():
    from speech_amd import ops
    src = rnd((70, 45), 1).float()
    dst = torch.zeros(45, 100, device=dev())
    ops.transpose(src.to(dev()), dst, 100, 20)
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu()[:, 20:90], src.t())
    assert dst.cpu()[:, :20].abs().sum() == 0


def test_transpose_batch():
    """Many matrices in one call (more than one launch's worth of table entries), two of them into one destination."""
    from speech_amd import ops
    shapes = [(70, 45), (128, 512), (512, 128), (33, 1), (1, 40)] * 14          # 70 items
    srcs = [rnd(sh, 100 + i).float().to(dev()) for i, sh in enumerate(shapes)]
    dsts = [torch.zeros(K, M + 7, device=dev()) for (M, K) in shapes]
    entries = [(s_, d_, d_.shape[1], 3) for s_, d_ in zip(srcs, dsts)]
    a, b = rnd((20, 64), 300).float().to(dev()), rnd((44, 64), 301).float().to(dev())
    both = torch.zeros(64, 64, device=dev())
    entries += [(a, both, 64, 0), (b, both, 64, 20)]
    ops.transpose_batch(ops.transpose_items(entries))
    torch.cuda.synchronize()
    for s_, d_ in zip(srcs, dsts):
        M = s_.shape[0]
        assert torch.equal(d_[:, 3:3 + M], s_.t())
        assert d_[:, :3].abs().sum() == 0 and d_[:, 3 + M:].abs().sum() == 0
    assert torch.equal(both, torch.cat([a, b], 0).t())


@pytest.mark.parametrize("B,F,S,K", [(2, 5, 7, 9), (3, 64, 33, 50), (1, 64, 31, 125), (2, 33, 1, 70)])
def test_dual_path_axis_shuffles(B, F, S, K):
    """ctn_permute through its five callers against torch's permute: bit-exact (copies, one addition)."""
    from speech_amd import ops
    x = rnd((B, F, S, K), 11).float().to(dev())
    for walk in (False, True):
        want = x.permute(0, 3, 2, 1) if walk else x.permute(0, 2, 3, 1)
        r = ops.rows_from_blocks(x, walk)
        assert r.is_contiguous() and torch.equal(r, want.contiguous())
        assert torch.equal(ops.blocks_from_rows(r, walk), x)
    rows = rnd((B, S * K, F), 12).float().to(dev())
    ff = ops.feature_first(rows)                                           # activation layout: rows padded to 16 bytes
    assert ops.is_act(ff) and torch.equal(ff, rows.permute(0, 2, 1))
    assert torch.equal(ops.feature_last(ff), rows)
    assert torch.equal(ops.feature_last(rows.permute(0, 2, 1).contiguous()), rows)
    z = ops.new_act(B, F, K * S, dev())
    z.copy_(rnd((B, F, K * S), 13).float())
    z4 = z.view(B, F, K, S)
    assert torch.equal(ops.swap_blocks(z4), z4.permute(0, 1, 3, 2).contiguous())
    assert torch.equal(ops.swap_blocks(z4, addend=x), z4.permute(0, 1, 3, 2) + x)
    back = ops.swap_blocks(x, act=True)                                    # (B, F, K, S) view of an activation
    assert ops.is_act(back.reshape(B, F, K * S)) and back.reshape(B, F, K * S).data_ptr() == back.data_ptr()
    assert torch.equal(back, x.permute(0, 1, 3, 2))
    with pytest.raises(AssertionError):
        ops.permute(x, torch.empty(4, device=dev()), F, K, S * K, F)       # strides beyond the destination


@pytest.mark.parametrize("M,K,T,batch", [(40, 24, 301, 2), (128, 128, 515, 2), (200, 130, 257, 1), (512, 128, 3999, 1),
                                         (32, 512, 100, 3), (64, 64, 64, 1)])
@pytest.mark.parametrize("width", [64, 128])
def test_gemm_nn_plain(M, K, T, batch, width):
    from speech_amd import ops
    W = rnd((M, K), 2)
    bias = rnd((M,), 3)
    X = rnd((batch, K, T), 4)
    ref = torch.einsum("mk,bkt->bmt", W, X) + bias.view(1, M, 1)
    Wt = torch.empty(K, M, device=dev())
    ops.transpose(W.float().to(dev()), Wt, M)
    Y = ops.new_act(batch, M, T, dev())
    ops.gemm_nn(M, T, K, batch, Wt, M, act(X), Y, bias=bias.float().to(dev()), tile_width=width)
    torch.cuda.synchronize()
    assert_close(Y, ref, TOL, "gemm_nn")


@pytest.mark.parametrize("gemm_path", PATHS, indirect=True)
def test_gemm_nn_split_k_add_residual(gemm_path):
    from speech_amd import ops
    M, K1, K2, T, B = (48, 16, 24, 301, 2) if gemm_path == "generic" else (72, 16, 32, 301, 2)
    Wa, Wb = rnd((K1, M), 5), rnd((K2, M), 6)       # already k-major
    Xa, Xb, R = rnd((B, K1, T), 7), rnd((B, K2, T), 8), rnd((B, M, T), 9)
    ref = torch.einsum("km,bkt->bmt", Wa, Xa) + torch.einsum("km,bkt->bmt", Wb, Xb) + R
    Y = ops.new_act(B, M, T, dev())
    ops.gemm_nn(M, T, K1 + K2, B, Wa.float().to(dev()), M, act(Xa), Y, A2=Wb.float().to(dev()), lda2=M, K1=K1,
                Xb=act(Xb), emode=ops.EPI_ADD, R=act(R))
    torch.cuda.synchronize()
    assert_close(Y, ref, TOL, "split-K + residual")


@pytest.mark.parametrize("gemm_path", PATHS, indirect=True)
@pytest.mark.parametrize("cln", [False, True])
def test_gemm_nn_norm_on_load_heads_epilogue(cln, gemm_path):
    from speech_amd import ops
    H, Bc, Sc, T, B = (40, 16, 24, 301, 2) if gemm_path == "generic" else (48, 72, 64, 301, 2)
    y2 = rnd((B, H, T), 10) + 0.3
    alpha = torch.tensor([0.2], dtype=torch.float64)
    gamma, beta = 1 + 0.2 * rnd((H,), 11), 0.2 * rnd((H,), 12)
    u = prelu(y2, alpha)
    mean, rstd = (cln_stats if cln else gln_stats)(u)
    n = (u - mean) * rstd * gamma.view(1, H, 1) + beta.view(1, H, 1)
    Wo, Ws, bo, bs = rnd((Bc, H), 13), rnd((Sc, H), 14), rnd((Bc,), 15), rnd((Sc,), 16)
    x, skip0 = rnd((B, Bc, T), 17), rnd((B, Sc, T), 18)
    ref_out = torch.einsum("mk,bkt->bmt", Wo, n) + bo.view(1, -1, 1) + x
    ref_skip = torch.einsum("mk,bkt->bmt", Ws, n) + bs.view(1, -1, 1)
    Wct = torch.empty(H, Bc + Sc, device=dev())
    ops.transpose(Wo.float().to(dev()), Wct, Bc + Sc, 0)
    ops.transpose(Ws.float().to(dev()), Wct, Bc + Sc, Bc)
    st = make_stat(mean, rstd, cln)
    f = lambda t: t.float().to(dev())
    for accumulate in (False, True):
        out = ops.new_act(B, Bc, T, dev())
        skip = act(skip0.clone())
        ops.gemm_nn(Bc + Sc, T, H, B, Wct, Bc + Sc, act(y2), out, xflags=ops.XF_PRELU | ops.XF_NORM, alpha=f(alpha),
                    stat=st, gamma=f(gamma), beta=f(beta), bias=f(bo), bias2=f(bs), emode=ops.EPI_HEADS, Mo=Bc,
                    R=act(x), Y2=skip, accumulate=accumulate)
        torch.cuda.synchronize()
        assert_close(out, ref_out, TOL, "heads out")
        assert_close(skip, ref_skip + (skip0 if accumulate else 0), TOL, "heads skip")


@pytest.mark.parametrize("gemm_path", PATHS, indirect=True)
@pytest.mark.parametrize("cln", [False, True])
def test_gemm_nn_stats_epilogue(cln, gemm_path):
    from speech_amd import ops
    M, K, T, B = (60, 20, 301, 2) if gemm_path == "generic" else (136, 32, 301, 2)
    W, bias, X = rnd((M, K), 20), rnd((M,), 21), rnd((B, K, T), 22)
    alpha = torch.tensor([0.25], dtype=torch.float64)
    ref = torch.einsum("mk,bkt->bmt", W, X) + bias.view(1, M, 1)
    mean, rstd = (cln_stats if cln else gln_stats)(prelu(ref, alpha))
    Wt = torch.empty(K, M, device=dev())
    ops.transpose(W.float().to(dev()), Wt, M)
    Y = ops.new_act(B, M, T, dev())
    acc = ops.Accum(B, T, cln, dev())
    ops.gemm_nn(M, T, K, B, Wt, M, act(X), Y, bias=bias.float().to(dev()), emode=ops.EPI_STATS,
                ealpha=alpha.float().to(dev()), eacc=acc)
    st = ops.finalize_stats(acc, B, M, T, 1e-12, dev())
    torch.cuda.synchronize()
    assert_close(Y, ref, TOL, "Y")
    if cln:
        assert_close(st.mean[:, :T], mean.reshape(B, T), TOL, "mean")
        assert_close(st.rstd[:, :T], rstd.reshape(B, T), 5e-5, "rstd")
    else:
        assert_close(st.mean, mean.reshape(B), TOL, "mean")
        assert_close(st.rstd, rstd.reshape(B), TOL, "rstd")


@pytest.mark.parametrize("gemm_path", PATHS, indirect=True)
def test_gemm_nn_sigmoid_and_prelu_bwd_epilogues(gemm_path):
    from speech_amd import ops
    M, K, T, B = (96, 24, 130, 2) if gemm_path == "generic" else (132, 32, 130, 2)
    W, bias, X = rnd((M, K), 30), rnd((M,), 31), rnd((B, K, T), 32)
    alpha = torch.tensor([0.3], dtype=torch.float64)
    f = lambda t: t.float().to(dev())
    Wt = torch.empty(K, M, device=dev())
    ops.transpose(f(W), Wt, M)
    Y = ops.new_act(B, M, T, dev())
    ops.gemm_nn(M, T, K, B, Wt, M, act(X), Y, xflags=ops.XF_PRELU, alpha=f(alpha), bias=f(bias),
                emode=ops.EPI_SIGMOID)
    ref = torch.sigmoid(torch.einsum("mk,bkt->bmt", W, prelu(X, alpha)) + bias.view(1, M, 1))
    torch.cuda.synchronize()
    assert_close(Y, ref, TOL, "sigmoid")
    # PReLU backward epilogue
    E = rnd((B, M, T), 33)
    acc = torch.einsum("mk,bkt->bmt", W, X)
    ref_y = torch.where(E > 0, acc, alpha * acc)
    ref_da = (acc * torch.where(E > 0, torch.zeros_like(E), E)).sum()
    da = torch.zeros(ops._lib.STAT_SLOTS, device=dev(), dtype=torch.float64)
    ops.gemm_nn(M, T, K, B, Wt, M, act(X), Y, emode=ops.EPI_PRELUBWD, E=act(E), ealpha=f(alpha), esums=da)
    torch.cuda.synchronize()
    assert_close(Y, ref_y, TOL, "prelu bwd")
    assert_close(da.sum().reshape(1), ref_da.reshape(1), 1e-4, "dalpha")


@pytest.mark.parametrize("gemm_path", PATHS, indirect=True)
@pytest.mark.parametrize("cln", [False, True])
def test_gemm_nn_normbwd_epilogue_and_load(cln, gemm_path):
    from speech_amd import ops
    H, K, T, B = (40, 24, 301, 2) if gemm_path == "generic" else (144, 32, 301, 2)
    f = lambda t: t.float().to(dev())
    W = rnd((K, H), 40)               # k-major already: dn = W^T-contract
    X = rnd((B, K, T), 41)
    y = rnd((B, H, T), 42) + 0.2
    alpha = torch.tensor([0.2], dtype=torch.float64)
    gamma = 1 + 0.2 * rnd((H,), 43)
    u = prelu(y, alpha)
    mean, rstd = (cln_stats if cln else gln_stats)(u)
    xhat = (u - mean) * rstd
    dn = torch.einsum("km,bkt->bmt", W, X)
    g = gamma.view(1, H, 1) * dn
    st = make_stat(mean, rstd, cln)
    Y = ops.new_act(B, H, T, dev())
    acc = ops.Accum(B, T, cln, dev())
    ops.gemm_nn(H, T, K, B, f(W), H, act(X), Y, emode=ops.EPI_NORMBWD, E=act(y), ealpha=f(alpha), estat=st,
                egamma=f(gamma), eacc=acc)
    ops.bwd_coefs(acc, st, B, H, T, 1e-12)
    torch.cuda.synchronize()
    assert_close(Y, dn, TOL, "dn")
    if cln:
        assert_close(acc.col[:, 0, :T], g.sum(1), 1e-4, "G1")
        assert_close(acc.col[:, 1, :T], (g * xhat).sum(1), 1e-4, "G2")
    else:
        assert_close(acc.sums.sum(1)[:, 0], g.sum((1, 2)), 1e-4, "S1")
        assert_close(acc.sums.sum(1)[:, 1], (g * xhat).sum((1, 2)), 1e-4, "S2")
    # reference du through autograd of the norm in float64
    yv = y.clone().requires_grad_(True)
    uu = prelu(yv, alpha)
    m2, r2 = (cln_stats if cln else gln_stats)(uu)
    nn_ = (uu - m2) * r2 * gamma.view(1, H, 1)
    nn_.backward(dn)
    dy_ref = yv.grad
    # NORMBWD on load feeding a second GEMM
    M2 = 24 if gemm_path == "generic" else 72
    W2 = rnd((H, M2), 44)
    R = rnd((B, M2, T), 45)
    ref2 = torch.einsum("km,bkt->bmt", W2, dy_ref) + R
    Y2 = ops.new_act(B, M2, T, dev())
    ops.gemm_nn(M2, T, H, B, f(W2), M2, Y, Y2, xflags=ops.XF_NORMBWD, X2=act(y), alpha=f(alpha), stat=st,
                gamma=f(gamma), emode=ops.EPI_ADD, R=act(R))
    torch.cuda.synchronize()
    assert_close(Y2, ref2, 1e-4, "normbwd on load")
    # the same transform inside the weight-gradient GEMM, with bias / slope reductions
    NB = {"generic": 20, "fast64": 80, "fast128": 128}[gemm_path]     # 128: the wgrad kernel's 64 x 128 tile
    Bm = rnd((B, NB, T), 46)
    dW = torch.zeros(H, NB, device=dev())
    db = torch.zeros(H, device=dev())
    da = torch.zeros(ops._lib.STAT_SLOTS, device=dev(), dtype=torch.float64)
    ops.gemm_nt(H, NB, T, B, Y, act(Bm), dW, NB, aflags=ops.XF_NORMBWD, A2=act(y), a_alpha=f(alpha), a_stat=st,
                a_gamma=f(gamma), a_rowsum=db, a_dalpha=da, kchunk=64)
    torch.cuda.synchronize()
    assert_close(dW, torch.einsum("bmt,bnt->mn", dy_ref, Bm), 1e-4, "wgrad normbwd")
    assert_close(db, dy_ref.sum((0, 2)), 1e-4, "db")
    du_ref = torch.where(y > 0, dy_ref, dy_ref / alpha)
    assert_close(da.sum().reshape(1), (du_ref * torch.where(y > 0, torch.zeros_like(y), y)).sum().reshape(1), 1e-4,
                 "dalpha")


@pytest.mark.parametrize("M,N,T,batch,kchunk", [(40, 24, 301, 2, 64), (128, 128, 515, 2, 256), (200, 72, 1000, 1, 512),
                                                (512, 128, 3999, 1, None), (144, 128, 301, 2, 64),
                                                (128, 256, 1021, 3, 256)])
def test_gemm_nt_plain(M, N, T, batch, kchunk):
    from speech_amd import ops
    A, Bm = rnd((batch, M, T), 50), rnd((batch, N, T), 51)
    ref = torch.einsum("bmt,bnt->mn", A, Bm)
    C = torch.zeros(M, N, device=dev())
    rs = torch.zeros(M, device=dev())
    ops.gemm_nt(M, N, T, batch, act(A), act(Bm), C, N, a_rowsum=rs, kchunk=kchunk)
    torch.cuda.synchronize()
    assert_close(C, ref, TOL, "gemm_nt")
    assert_close(rs, A.sum((0, 2)), TOL, "rowsum")


@pytest.mark.parametrize("cln", [False, True])
@pytest.mark.parametrize("fast", [False, True])
def test_gemm_nt_split_rows_norm_b(fast, cln):
    from speech_amd import ops
    M1, M2, N, T, B = (80, 64, 128, 301, 2) if fast else (16, 24, 40, 301, 2)   # fast: the LDS-DMA kernel's shapes
    f = lambda t: t.float().to(dev())
    A1, A2 = rnd((B, M1, T), 52), rnd((B, M2, T), 53)
    y = rnd((B, N, T), 54) + 0.1
    alpha = torch.tensor([0.2], dtype=torch.float64)
    gamma, beta = 1 + 0.2 * rnd((N,), 55), 0.2 * rnd((N,), 56)
    u = prelu(y, alpha)
    mean, rstd = (cln_stats if cln else gln_stats)(u)
    n = (u - mean) * rstd * gamma.view(1, N, 1) + beta.view(1, N, 1)
    C1 = torch.zeros(M1, N, device=dev())
    C2 = torch.zeros(M2, N, device=dev())
    r1 = torch.zeros(M1, device=dev())
    r2 = torch.zeros(M2, device=dev())
    ops.gemm_nt(M1 + M2, N, T, B, act(A1), act(y), C1, N, Ab=act(A2), M1=M1, C2=C2, ldc2=N, a_rowsum=r1, a_rowsum2=r2,
                bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=f(alpha), b_stat=make_stat(mean, rstd, cln),
                b_gamma=f(gamma), b_beta=f(beta), kchunk=128)
    torch.cuda.synchronize()
    assert_close(C1, torch.einsum("bmt,bnt->mn", A1, n), TOL, "C1")
    assert_close(C2, torch.einsum("bmt,bnt->mn", A2, n), TOL, "C2")
    assert_close(r1, A1.sum((0, 2)), TOL, "r1")
    assert_close(r2, A2.sum((0, 2)), TOL, "r2")


@pytest.mark.parametrize("d,P,pad_left,T", [(1, 3, 1, 301), (2, 3, 4, 301), (128, 3, 128, 3999), (128, 3, 256, 1500),
                                            (3, 2, 0, 77), (16, 5, 32, 9000)])
def test_depthwise_plain(d, P, pad_left, T):
    import torch.nn.functional as F
    from speech_amd.functional import DepthwiseFn
    C, B = 12, 2
    x = rnd((B, C, T), 60).requires_grad_(True)
    w = rnd((C, 1, P), 61).requires_grad_(True)
    b = rnd((C,), 62).requires_grad_(True)
    pad_right = (P - 1) * d - pad_left
    ref = F.conv1d(F.pad(x, (pad_left, pad_right)), w, b, dilation=d, groups=C)
    gy = rnd(tuple(ref.shape), 63)
    ref.backward(gy)
    xg = x.detach().float().to(dev()).requires_grad_(True)
    wg = w.detach().float().to(dev()).requires_grad_(True)
    bg = b.detach().float().to(dev()).requires_grad_(True)
    y = DepthwiseFn.apply(xg, wg, bg, d, pad_left, pad_right)
    y.backward(gy.float().to(dev()))
    torch.cuda.synchronize()
    assert_close(y, ref, TOL, "dw y")
    assert_close(xg.grad, x.grad, TOL, "dw dx")
    assert_close(wg.grad, w.grad, 5e-5, "dw dw")
    assert_close(bg.grad, b.grad, 5e-5, "dw db")


@pytest.mark.parametrize("d,pad_left,T", [(1, 1, 3999), (2, 2, 3999), (8, 16, 3999), (128, 128, 3999), (64, 128, 4096),
                                          (3, 3, 1001), (1, 0, 5)])
def test_depthwise_backward_recomputed_output_matches_stored(d, pad_left, T):
    """The whole-row gLN backward kernel in its two forms -- reading the stored convolution output, and recomputing it
    from the input row (three tensors over HBM instead of four) -- on the forward kernel's own output and statistics:
    every gradient agrees to the kernel bar.  (A recomputed pre-activation can differ from the stored one only if the
    two kernels round differently; they are written expression for expression alike.)"""
    from speech_amd import ops
    B, H, P = 3, 20, 3
    x = act(rnd((B, H, T), 70))
    dn2 = act(rnd((B, H, T), 71))
    w, bias = rnd((H, P), 72).float().to(dev()), rnd((H,), 73).float().to(dev())
    g1, be1, g2 = (rnd((H,), 74 + i).float().to(dev()) for i in range(3))
    a1 = torch.tensor([0.25], device=dev())
    a2 = torch.tensor([0.1], device=dev())
    u1 = prelu(x.double().cpu(), 0.25)
    st1 = make_stat(*gln_stats(u1), False)
    y2 = ops.new_act(B, H, T, dev())
    acc2 = ops.Accum(B, T, False, dev())
    ops.dw_fwd(x, w, bias, y2, d, pad_left, xflags=ops.XF_PRELU | ops.XF_NORM, alpha_in=a1, stat=st1, gamma=g1, beta=be1,
               alpha_out=a2, acc=acc2)
    st2 = ops.finalize_stats(acc2, B, H, T, 1e-12, dev())
    st2.c1 = rnd((B,), 80).float().to(dev()) * 0.01
    st2.c2 = rnd((B,), 81).float().to(dev()) * 0.01
    outs = []
    for recompute in (False, True):
        dw, db = torch.zeros(H, P, device=dev()), torch.zeros(H, device=dev())
        dg1, dbe1, dg2, dbe2 = (torch.zeros(H, device=dev()) for _ in range(4))
        da2 = torch.zeros(ops._lib.STAT_SLOTS, device=dev(), dtype=torch.float64)
        iacc = ops.Accum(B, T, False, dev())
        dx = ops.new_act(B, H, T, dev())
        ops.dw_bwd(dn2, x, w, dw, db, dx, d, pad_left, xflags=ops.XF_PRELU | ops.XF_NORM, alpha1=a1, stat1=st1, gamma1=g1,
                   beta1=be1, dgamma1=dg1, dbeta1=dbe1, iacc=iacc, y2=y2, alpha2=a2, stat2=st2, gamma2=g2, dgamma2=dg2,
                   dbeta2=dbe2, dalpha2=da2, bias=bias, recompute_y2=recompute)
        torch.cuda.synchronize()
        outs.append(dict(dx=dx.clone(), dw=dw, db=db, dg1=dg1, dbe1=dbe1, dg2=dg2, dbe2=dbe2, da2=da2.sum().reshape(1),
                         isums=iacc.sums.sum(dim=1)))
    for k in outs[0]:
        assert_close(outs[1][k], outs[0][k], TOL, "recomputed vs stored: " + k)


@pytest.mark.parametrize("L,S,T,relu", [(16, 8, 1000, False), (16, 8, 32000, True), (2, 1, 500, False), (16, 4, 403, True),
                                        (16, 8, 16, False),       # one frame
                                        (16, 16, 4096, True),     # no overlap between frames
                                        (10, 5, 1005, False),     # taps and stride that are not powers of two
                                        (3, 1, 77, True), (8, 2, 64001, False), (2, 1, 64000, False),
                                        (20, 10, 4410, False),    # more than 16 taps (the 44.1 kHz music recipes use L = 20)
                                        (32, 16, 2048, True), (24, 6, 999, False), (17, 1, 300, True)])
def test_encoder_decoder(L, S, T, relu):
    import convtasnet_oracle as O
    from speech_amd.functional import EncoderFn, DecoderFn
    N, B = 48, 2
    T = (T - L) // S * S + L
    x = rnd((B, 1, T), 70).requires_grad_(True)
    We = (rnd((N, 1, L), 71) * 0.3).requires_grad_(True)
    Wd = (rnd((N, 1, L), 72) * 0.3).requires_grad_(True)
    w = O.encoder(x, We, S, "relu" if relu else None)
    xr = O.decoder(w, Wd, S)
    gx = rnd(tuple(xr.shape), 73)
    xr.backward(gx)
    xg = x.detach().float().to(dev()).requires_grad_(True)
    Weg = We.detach().float().to(dev()).requires_grad_(True)
    Wdg = Wd.detach().float().to(dev()).requires_grad_(True)
    wg = EncoderFn.apply(xg, Weg, S, relu)
    xrg = DecoderFn.apply(wg, Wdg, S)
    xrg.backward(gx.float().to(dev()))
    torch.cuda.synchronize()
    assert_close(wg, w, TOL, "encoder")
    assert_close(xrg, xr, TOL, "decoder")
    assert_close(Wdg.grad, Wd.grad, 1e-4, "dWd")
    assert_close(Weg.grad, We.grad, 1e-4, "dWe")
    assert_close(xg.grad, x.grad, 1e-4, "dx")


def test_adam_and_clip():
    import convtasnet_oracle as O
    from speech_amd import ops
    n = 100003
    p = rnd((n,), 80).float()
    sd = {"p": p.clone()}
    opt = O.AdamState(sd)
    pg = p.clone().to(dev())
    m = torch.zeros(n, device=dev())
    v = torch.zeros(n, device=dev())
    for step in range(1, 4):
        g = (rnd((n,), 80 + step) * (3.0 if step == 2 else 0.001)).float()
        gl = [g.clone()]
        O.clip_grad_norm_(gl, 5.0)
        opt.step(sd, {"p": gl[0]})
        acc = torch.zeros(1, device=dev(), dtype=torch.float64)
        gg = g.to(dev())
        ops.sumsq(gg, acc)
        ops.adam_step(pg, gg, m, v, acc, 5.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, step)
        torch.cuda.synchronize()
        assert_close(acc, (g.double() ** 2).sum().reshape(1), 1e-6, "sumsq")
        assert_close(pg, sd["p"], 1e-6, "adam step {}".format(step))
