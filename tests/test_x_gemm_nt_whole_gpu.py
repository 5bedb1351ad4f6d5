"""The whole-output weight-gradient kernels (csrc/ctn_gemm_nt_whole.hip, reached through ctn_gemm_nt at the residual
layer's two shapes when CTN_NT_WHOLE=1 is set and the launch has enough stages) against float64: the heads' form (two A tensors, B with
PReLU + norm on load, both bias gradients) and the bottleneck's form (norm-backward A operand with its bias and
PReLU-slope reductions), ragged frame counts, several batch elements, a single head, strided operand views."""
import pytest
import torch

from conftest import assert_close
from test_kernels_gpu import rnd, act, prelu, gln_stats, make_stat, dev, TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def whole_output_kernels(monkeypatch):
    """Switches the kernels on for the test and checks afterwards that they, not the tile kernels, ran."""
    from speech_amd import _lib
    monkeypatch.setenv("CTN_NT_WHOLE", "1")
    before = _lib.load().ctn_gemm_nt_whole_launches()
    yield
    assert _lib.load().ctn_gemm_nt_whole_launches() > before, "the call fell through to the tile kernels"


@pytest.mark.parametrize("M1,M2,N,T,B", [(128, 128, 512, 3999, 5), (128, 128, 256, 8191, 4), (256, 0, 512, 4100, 4),
                                          (128, 128, 512, 4001, 5), (1024, 0, 128, 3999, 6)])
def test_heads_form(M1, M2, N, T, B):
    from speech_amd import ops
    f = lambda t: t.float().to(dev())
    A1 = rnd((B, M1, T), 152)
    A2 = rnd((B, M2, T), 153) if M2 else None
    y = rnd((B, N, T), 154) + 0.1
    alpha = torch.tensor([0.2], dtype=torch.float64)
    gamma, beta = 1 + 0.2 * rnd((N,), 155), 0.2 * rnd((N,), 156)
    u = prelu(y, alpha)
    mean, rstd = gln_stats(u)
    n = (u - mean) * rstd * gamma.view(1, N, 1) + beta.view(1, N, 1)
    C1 = torch.zeros(M1, N, device=dev())
    C2 = torch.zeros(M2, N, device=dev()) if M2 else None
    r1 = torch.zeros(M1, device=dev())
    r2 = torch.zeros(M2, device=dev()) if M2 else None
    ops.gemm_nt(M1 + M2, N, T, B, act(A1), act(y), C1, N, Ab=act(A2) if M2 else None, M1=M1, C2=C2, ldc2=N if M2 else 0,
                a_rowsum=r1, a_rowsum2=r2, bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=f(alpha),
                b_stat=make_stat(mean, rstd, False), b_gamma=f(gamma), b_beta=f(beta))
    torch.cuda.synchronize()
    assert_close(C1, torch.einsum("bmt,bnt->mn", A1, n), TOL, "C1")
    assert_close(r1, A1.sum((0, 2)), TOL, "r1")
    if M2:
        assert_close(C2, torch.einsum("bmt,bnt->mn", A2, n), TOL, "C2")
        assert_close(r2, A2.sum((0, 2)), TOL, "r2")


@pytest.mark.parametrize("H,NB,T,B", [(512, 128, 3999, 5), (256, 128, 5000, 7), (512, 128, 4001, 5)])
def test_bottleneck_form(H, NB, T, B):
    from speech_amd import ops
    f = lambda t: t.float().to(dev())
    dn = rnd((B, H, T), 160)
    y = rnd((B, H, T), 161) + 0.2
    Bm = rnd((B, NB, T), 162)
    alpha = torch.tensor([0.2], dtype=torch.float64)
    gamma = 1 + 0.2 * rnd((H,), 163)
    yv = y.clone().requires_grad_(True)
    uu = prelu(yv, alpha)
    m2, r2 = gln_stats(uu)
    ((uu - m2) * r2 * gamma.view(1, H, 1)).backward(dn)
    dy_ref = yv.grad
    # the backward coefficients of the norm from float64 sums, as ctn_gln_bwd_coefs forms them
    u = prelu(y, alpha)
    mean, rstd = gln_stats(u)
    xhat = (u - mean) * rstd
    g = gamma.view(1, H, 1) * dn
    n = float(H * T)
    st = make_stat(mean, rstd, False)
    st.c1 = f((-rstd.view(B) * g.sum((1, 2)) / n))
    st.c2 = f((-rstd.view(B) ** 2 * (g * xhat).sum((1, 2)) / n))
    dW = torch.zeros(H, NB, device=dev())
    db = torch.zeros(H, device=dev())
    da = torch.zeros(ops._lib.STAT_SLOTS, device=dev(), dtype=torch.float64)
    ops.gemm_nt(H, NB, T, B, act(dn), act(Bm), dW, NB, aflags=ops.XF_NORMBWD, A2=act(y), a_alpha=f(alpha), a_stat=st,
                a_gamma=f(gamma), a_rowsum=db, a_dalpha=da)
    torch.cuda.synchronize()
    assert_close(dW, torch.einsum("bmt,bnt->mn", dy_ref, Bm), 1e-4, "wgrad normbwd")
    assert_close(db, dy_ref.sum((0, 2)), 1e-4, "db")
    du_ref = torch.where(y > 0, dy_ref, dy_ref / alpha)
    assert_close(da.sum().reshape(1), (du_ref * torch.where(y > 0, torch.zeros_like(y), y)).sum().reshape(1), 1e-4,
                 "dalpha")


def test_strided_views_and_accumulation():
    """Operands as views of larger buffers (row pitch beyond T, batch stride beyond the tensor), and a second call adding
    to the first one's output."""
    from speech_amd import ops
    M1, M2, N, T, B = 128, 128, 512, 3000, 6
    ld = 3008
    bufA = torch.randn(B, M1 + M2 + 8, ld, device=dev())
    bufY = torch.randn(B, N + 4, ld, device=dev())
    A1, A2, y = bufA[:, :M1, :T], bufA[:, M1:M1 + M2, :T], bufY[:, :N, :T]
    C1 = torch.zeros(M1, N, device=dev())
    C2 = torch.zeros(M2, N, device=dev())
    for _ in range(2):
        ops.gemm_nt(M1 + M2, N, T, B, A1, y, C1, N, Ab=A2, M1=M1, C2=C2, ldc2=N)
    torch.cuda.synchronize()
    A1c, A2c, yc = A1.double().cpu(), A2.double().cpu(), y.double().cpu()
    assert_close(C1, 2 * torch.einsum("bmt,bnt->mn", A1c, yc), TOL, "C1")
    assert_close(C2, 2 * torch.einsum("bmt,bnt->mn", A2c, yc), TOL, "C2")


def test_partials_through_the_workspace_are_added_in_a_fixed_order(monkeypatch):
    """With a workspace the blocks store their partial outputs and a second launch adds them in block order: two runs
    give the same bits (the atomics route differs from run to run in the last place), and both routes agree."""
    from speech_amd import ops
    M1 = M2 = 128
    N, T, B = 512, 3999, 6
    f = lambda t: t.float().to(dev())
    A1, A2, y = act(rnd((B, M1, T), 170)), act(rnd((B, M2, T), 171)), act(rnd((B, N, T), 172) + 0.1)
    alpha = f(torch.tensor([0.2], dtype=torch.float64))
    gamma, beta = f(1 + 0.2 * rnd((N,), 173)), f(0.2 * rnd((N,), 174))
    st = make_stat(torch.full((B, 1, 1), 0.05, dtype=torch.float64), torch.full((B, 1, 1), 1.3, dtype=torch.float64), False)

    def run():
        C1, C2 = torch.zeros(M1, N, device=dev()), torch.zeros(M2, N, device=dev())
        r1, r2 = torch.zeros(M1, device=dev()), torch.zeros(M2, device=dev())
        ops.gemm_nt(M1 + M2, N, T, B, A1, y, C1, N, Ab=A2, M1=M1, C2=C2, ldc2=N, a_rowsum=r1, a_rowsum2=r2,
                    bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=alpha, b_stat=st, b_gamma=gamma, b_beta=beta)
        torch.cuda.synchronize()
        return C1, C2, r1, r2

    first, second = run(), run()
    for x, y_ in zip(first, second):
        assert torch.equal(x, y_)
    monkeypatch.setattr(ops, "_nt_workspace", lambda *a: None)
    third = run()
    for x, y_ in zip(first, third):
        assert_close(x, y_.double().cpu(), TOL, "atomics route")


def test_bottleneck_form_is_reproducible():
    """Two runs of the norm-backward form through the workspace give the same bits in the weight and bias gradients
    (the PReLU-slope gradient goes through fp64 atomics and may differ in its last place)."""
    from speech_amd import ops
    H, NB, T, B = 512, 128, 3999, 6
    f = lambda t: t.float().to(dev())
    dn, y, Bm = act(rnd((B, H, T), 180)), act(rnd((B, H, T), 181) + 0.2), act(rnd((B, NB, T), 182))
    alpha, gamma = f(torch.tensor([0.2], dtype=torch.float64)), f(1 + 0.2 * rnd((H,), 183))
    st = make_stat(0.05 * rnd((B, 1, 1), 184), 1.0 + 0.1 * rnd((B, 1, 1), 185).abs(), False)
    st.c1, st.c2 = f(1e-3 * rnd((B,), 186)), f(1e-3 * rnd((B,), 187))

    def run():
        dW, db = torch.zeros(H, NB, device=dev()), torch.zeros(H, device=dev())
        da = torch.zeros(ops._lib.STAT_SLOTS, device=dev(), dtype=torch.float64)
        ops.gemm_nt(H, NB, T, B, dn, Bm, dW, NB, aflags=ops.XF_NORMBWD, A2=y, a_alpha=alpha, a_stat=st, a_gamma=gamma,
                    a_rowsum=db, a_dalpha=da)
        torch.cuda.synchronize()
        return dW, db, da.sum()

    runs = [run() for _ in range(3)]
    for dW, db, da in runs[1:]:
        assert torch.equal(dW, runs[0][0]) and torch.equal(db, runs[0][1])
        assert abs(float(da - runs[0][2])) <= 1e-9 * abs(float(runs[0][2])) + 1e-12
