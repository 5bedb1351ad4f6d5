"""The fused backward pass of a layer's pointwise heads (ctn_heads_bwd: input gradient with its norm sums, weight and
bias gradients in one launch) against float64 math and against the two launches it replaces (ctn_gemm_nn with the
norm-backward epilogue, ctn_gemm_nt), on the paper's head shapes and on ragged frame counts, one and two heads."""
import pytest
import torch

from conftest import assert_close
from speech_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(B, T, H, M1, M2, seed):
    from speech_amd import ops
    rnd = lambda shape, k: synth.normal(shape, seed + k)
    y2 = rnd((B, H, T), 1)
    dY1 = rnd((B, M1, T), 2) if M1 else None
    dY2 = rnd((B, M2, T), 3)
    W1 = rnd((M1, H), 4) * 0.1 if M1 else None
    W2 = rnd((M2, H), 5) * 0.1
    gamma, beta = rnd((H,), 6), rnd((H,), 7) * 0.3
    alpha = 0.25
    # float64 reference
    y64 = y2.double()
    u = torch.where(y64 > 0, y64, alpha * y64)
    mean = u.mean(dim=(1, 2), keepdim=True)
    var = u.var(dim=(1, 2), unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-12)
    xhat = (u - mean) * rstd
    n2 = xhat * gamma.double().view(1, H, 1) + beta.double().view(1, H, 1)
    dY = torch.cat(([dY1.double()] if M1 else []) + [dY2.double()], dim=1)
    Wh = torch.cat(([W1.double()] if M1 else []) + [W2.double()], dim=0)
    ref = dict(dn2=torch.einsum("mc,bmt->bct", Wh, dY))
    g = gamma.double().view(1, H, 1)
    ref["S1"] = (g * ref["dn2"]).sum(dim=(1, 2))
    ref["S2"] = (g * ref["dn2"] * xhat).sum(dim=(1, 2))
    ref["dW"] = torch.einsum("bmt,bct->mc", dY, n2)
    ref["db"] = dY.sum(dim=(0, 2))
    # GPU
    act = lambda t: ops.as_act(t.to(DEV))
    st = ops.Stat(mean.view(B).float().to(DEV), rstd.view(B).float().to(DEV), 1, 0, False)
    parts = ([(W1.to(DEV), False, 0, 0)] if M1 else []) + [(W2.to(DEV), False, M1, 0)]
    Wp = ops.prepared(M1 + M2, H, parts, DEV)
    dn2 = ops.new_act(B, H, T, DEV)
    eacc = ops.Accum(B, T, False, DEV)
    dW1 = torch.zeros(M1, H, device=DEV) if M1 else None
    dW2 = torch.zeros(M2, H, device=DEV)
    rs1 = torch.zeros(M1, device=DEV) if M1 else None
    rs2 = torch.zeros(M2, device=DEV)
    a_t = torch.tensor([alpha], device=DEV)
    assert ops.heads_bwd_usable(M1, M2, H, T, st)
    ops.heads_bwd(act(dY1) if M1 else None, act(dY2), act(y2), a_t, st, gamma.to(DEV), beta.to(DEV), Wp, dn2, eacc, dW1, dW2,
                  rs1, rs2)
    torch.cuda.synchronize()
    got = dict(dn2=dn2, S1=eacc.sums[:, :, 0].sum(dim=1), S2=eacc.sums[:, :, 1].sum(dim=1),
               dW=torch.cat(([dW1] if M1 else []) + [dW2], dim=0), db=torch.cat(([rs1] if M1 else []) + [rs2]))
    return got, ref


@pytest.mark.parametrize("B,T,H,M1,M2", [(2, 3999, 512, 128, 128), (3, 64, 128, 128, 128), (1, 5, 256, 0, 128),
                                         (2, 1001, 512, 0, 128), (5, 130, 128, 64, 64), (1, 4096, 256, 96, 160),
                                         (1, 1021, 512, 128, 128)])
def test_heads_bwd_matches_float64(B, T, H, M1, M2):
    got, ref = _case(B, T, H, M1, M2, 900)
    assert_close(got["dn2"], ref["dn2"], 2e-5, "dn2")
    assert_close(got["dW"], ref["dW"], 2e-5, "dW")
    assert_close(got["db"], ref["db"], 2e-5, "db")
    # the sums are dot products of dn2 with O(1) factors: held relative to the sum of the absolute terms
    scale1 = float((ref["dn2"].abs().sum(dim=(1, 2))).max())
    assert float((got["S1"].cpu().double() - ref["S1"]).abs().max()) <= 2e-5 * scale1
    assert float((got["S2"].cpu().double() - ref["S2"]).abs().max()) <= 2e-5 * scale1


def _model_grads(cfg, sd, mixture, sources, fused):
    from speech_amd import functional
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from test_parity_gpu import build, load_state, dev
    before = functional.FUSED_HEADS_BWD
    functional.FUSED_HEADS_BWD = fused
    try:
        model = load_state(build(cfg), sd)
        out = model(mixture.to(dev()))
        loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev()))
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach(), perm, {k: p.grad.clone() for k, p in model.named_parameters()}
    finally:
        functional.FUSED_HEADS_BWD = before


def test_training_path_with_the_fused_launch_agrees_with_the_two_launches():
    """functional.FUSED_HEADS_BWD routes every layer of the paper-width network (H = 512, two heads of 128 rows) through
    ctn_heads_bwd; loss, permutation and every gradient must agree with the default route (same arithmetic, another
    summation order: tensors within 2e-5 of their largest element, the scalar slope gradients within 2e-3)."""
    import cases
    import convtasnet_oracle as O
    cfg = dict(O.CONFIG_PAPER, sep_num_blocks=1, sep_num_layers=4)
    sd = cases.e2e_state(cfg)
    mixture, sources = cases.e2e_batch(cfg, 2, 8000, 3)
    la, pa, ga = _model_grads(cfg, sd, mixture, sources, True)
    lb, pb, gb = _model_grads(cfg, sd, mixture, sources, False)
    assert torch.equal(pa, pb)
    assert_close(la, lb, 1e-6, "loss")
    for k in gb:
        assert_close(ga[k], gb[k], 2e-3 if gb[k].numel() == 1 else 2e-5, "grad " + k)
