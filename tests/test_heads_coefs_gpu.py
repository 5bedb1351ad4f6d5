"""The norm-2 backward coefficients of a residual layer derived from by-products of the heads' weight-gradient launch
(ctn_heads_bwd_coefs, functional.HEADS_COEFS_FROM_WGRAD) against the statistics epilogue of the heads' input-gradient
GEMM: same loss, tensor gradients within 2e-5 of each tensor's largest element, scalar slope gradients (cancellation-heavy
sums, see functional.py) within 1e-2; and the by-products themselves against float64 math."""
import pytest
import torch

import cases
import convtasnet_oracle as O
from conftest import assert_close
from test_parity_gpu import build, load_state, dev

pytestmark = pytest.mark.gpu


def _grads(cfg, sd, mixture, sources, route):
    from speech_amd import functional
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    before = functional.HEADS_COEFS_FROM_WGRAD
    functional.HEADS_COEFS_FROM_WGRAD = route
    try:
        model = load_state(build(cfg), sd)
        out = model(mixture.to(dev()))
        loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev()))
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach(), perm, {k: p.grad.clone() for k, p in model.named_parameters()}
    finally:
        functional.HEADS_COEFS_FROM_WGRAD = before


@pytest.mark.parametrize("name,cfg,T", [("small", O.CONFIG_SMALL, 16000),
                                        ("paper", dict(O.CONFIG_PAPER, sep_num_blocks=1, sep_num_layers=4), 8000)])
def test_routes_agree(name, cfg, T):
    sd = cases.e2e_state(cfg)
    mixture, sources = cases.e2e_batch(cfg, 2, T, 3)
    la, pa, ga = _grads(cfg, sd, mixture, sources, True)
    lb, pb, gb = _grads(cfg, sd, mixture, sources, False)
    assert torch.equal(pa, pb)
    assert_close(la, lb, 1e-6, "loss")
    for k in gb:
        assert_close(ga[k], gb[k], 1e-2 if gb[k].numel() == 1 else 2e-5, "grad " + k)


def test_by_products_of_the_weight_gradient_launch():
    """a_rowsum_b and wdot_sums of ctn_gemm_nt, and the coefficients ctn_heads_bwd_coefs makes of them, against float64."""
    from speech_amd import ops
    from speech_amd.utils import synth
    B, Bc, Sc, H, T = 3, 32, 48, 64, 1500
    f64 = lambda shape, seed: synth.normal(shape, seed).double()
    dYo, dYs, y2 = f64((B, Bc, T), 1), f64((B, Sc, T), 2), f64((B, H, T), 3) + 0.2
    Wo, Ws = f64((Bc, H), 4), f64((Sc, H), 5)
    gamma, beta, alpha = 1 + 0.2 * f64((H,), 6), 0.3 * f64((H,), 7), torch.tensor([0.25], dtype=torch.float64)
    u = torch.where(y2 > 0, y2, alpha * y2)
    mean = u.mean(dim=(1, 2), keepdim=True)
    rstd = 1.0 / torch.sqrt(u.var(dim=(1, 2), unbiased=False, keepdim=True) + 1e-12)
    xhat = (u - mean) * rstd
    dY = torch.cat([dYo, dYs], 1)
    W = torch.cat([Wo, Ws], 0)
    dn = torch.einsum("mc,bmt->bct", W, dY)
    S1 = (gamma.view(1, H, 1) * dn).sum(dim=(1, 2))
    S2 = (gamma.view(1, H, 1) * dn * xhat).sum(dim=(1, 2))
    n = float(H * T)
    c1_ref, c2_ref = -rstd.view(B) * S1 / n, -rstd.view(B) ** 2 * S2 / n
    dW_ref = torch.einsum("bmt,bct->mc", dY, gamma.view(1, H, 1) * xhat + beta.view(1, H, 1))
    d = dev()
    f = lambda t: t.float().to(d)
    act = lambda t: ops.as_act(t.float().to(d))
    st = ops.Stat(f(mean.view(B)), f(rstd.view(B)), 1, 0, False)
    rs_b = torch.zeros(B, Bc + Sc, device=d)
    wsum = torch.zeros(B, ops._lib.STAT_SLOTS, device=d, dtype=torch.float64)
    dWo, dWs = torch.zeros(Bc, H, device=d), torch.zeros(Sc, H, device=d)
    bo, bs = torch.zeros(Bc, device=d), torch.zeros(Sc, device=d)
    Wod, Wsd = f(Wo), f(Ws)
    ops.gemm_nt(Bc + Sc, H, T, B, act(dYo), act(y2), dWo, H, Ab=act(dYs), M1=Bc, C2=dWs, ldc2=H, a_rowsum=bo, a_rowsum2=bs,
                bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=f(alpha), b_stat=st, b_gamma=f(gamma),
                b_beta=torch.zeros(H, device=d), a_rowsum_b=rs_b, wdot=Wod, wdot2=Wsd, wdot_sums=wsum)
    ops.heads_bwd_coefs(Wod, Wsd, f(gamma), f(beta), rs_b, wsum, st, B, H, T, dWo, dWs)
    torch.cuda.synchronize()
    assert_close(rs_b, dY.sum(-1), 2e-5, "per-sample row sums")
    assert_close(wsum.sum(1), S2, 1e-4, "sum W * G'")
    assert_close(st.c1, c1_ref, 1e-4, "c1")
    assert_close(st.c2, c2_ref, 1e-4, "c2")
    assert_close(torch.cat([dWo, dWs], 0), dW_ref, 2e-5, "weight gradient with its beta term restored")
    assert_close(torch.cat([bo, bs]), dY.sum(dim=(0, 2)), 2e-5, "bias gradients")
