"""The arithmetic modes of the GEMMs (CTN_ARITH_* in include/ctn.h) against each other and against float64.  The default
-- three bf16 pieces per fp32 operand, an exact representation, six products on the bf16 MFMA with fp32 accumulation --
must meet the bounds of plain f32 MFMAs: GEMMs within 2e-5 of float64 math, gradients within 1e-4 (scalars 1e-3) of the
f32-MFMA mode; every other GPU test of the suite runs in this default mode too.  Two pieces / three products is the opt-in
throughput mode: every GEMM within 5e-5 of float64 math, waveforms and loss within 1e-4 of the reference's outputs (golden
fixtures), permutations bit-exact, training trajectories; what it loosens is documented by the bounds below: gradients
agree with the f32-MFMA mode to 1e-3 of each tensor's largest entry (scalar PReLU-slope gradients, sums with heavy
cancellation, to 1e-2)."""
import numpy as np
import pytest
import torch

import cases
import convtasnet_oracle as O
from conftest import assert_close, rel_l2
from test_parity_gpu import build, load_state, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 2, 3], ids=["f32_mfma", "two_pieces", "three_pieces"])
def split_mode(request):
    from speech_amd import ops
    default = ops.get_gemm_split()
    ops.set_gemm_split(request.param)
    assert ops.get_gemm_split() == request.param
    try:
        yield request.param
    finally:
        ops.set_gemm_split(default)


def run(cfg, batch, T, seed):
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    model = load_state(build(cfg), cases.e2e_state(cfg))
    mixture, sources = cases.e2e_batch(cfg, batch, T, seed)
    out = model(mixture.to(dev()))
    loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev()))
    loss.backward()
    torch.cuda.synchronize()
    return out.detach(), loss.detach(), perm, {k: p.grad.clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize("M,K,T,batch", [(512, 128, 3999, 2), (128, 512, 3999, 2), (256, 512, 1000, 3), (144, 272, 515, 1)])
def test_split_gemms_vs_float64(split_mode, M, K, T, batch):
    from speech_amd import ops
    g = torch.Generator().manual_seed(M + K)
    W = torch.randn(M, K, generator=g, dtype=torch.float64)
    X = torch.randn(batch, K, T, generator=g, dtype=torch.float64)
    Xa = ops.new_act(batch, K, T, dev()); Xa.copy_(X.float())
    Y = ops.new_act(batch, M, T, dev())
    ops.gemm_nn(M, T, K, batch, W.t().contiguous().float().to(dev()), M, Xa, Y)
    tol = 5e-5 if split_mode == 2 else 2e-5
    assert_close(Y, torch.einsum("mk,bkt->bmt", W, X), tol, "pointwise GEMM")
    dY = torch.randn(batch, M, T, generator=g, dtype=torch.float64)
    dYa = ops.new_act(batch, M, T, dev()); dYa.copy_(dY.float())
    dW = torch.zeros(M, K, device=dev())
    ops.gemm_nt(M, K, T, batch, dYa, Xa, dW, K)
    torch.cuda.synchronize()
    assert_close(dW, torch.einsum("bmt,bkt->mk", dY, X), tol, "weight-gradient GEMM")


@pytest.mark.parametrize("name,cfg", [("e2e_small", O.CONFIG_SMALL), ("e2e_small_causal3", dict(O.CONFIG_SMALL, causal=True, n_sources=3, enc_nonlinear="relu")),
                                      ("e2e_paper", O.CONFIG_PAPER), ("e2e_paper_causal3", O.CONFIG_CAUSAL3)])
def test_split_mode_meets_the_end_to_end_bar(golden, split_mode, name, cfg):
    g = golden(name)
    out, loss, perm, _ = run(cfg, int(g["batch"]), int(g["T"]), 0)
    assert np.array_equal(perm.cpu().numpy(), g["perm"])
    assert_close(loss, g["loss"], 1e-4, "loss")
    assert_close(out[..., ::8], g["out"], 1e-4, "waveforms")


@pytest.mark.parametrize("pieces,tol_vec,tol_scalar", [(2, 1e-3, 1e-2), (3, 1e-4, 1e-3)])
def test_split_mode_gradients_close_to_f32_mode(pieces, tol_vec, tol_scalar):
    """Paper configuration with every PReLU slope at 1 (no kinks, so the comparison sees the arithmetic and not which side
    of zero a pre-activation fell): gradients of a split mode and the f32-MFMA mode agree to tol_vec of each tensor's
    largest entry, scalar slope gradients (sums with heavy cancellation) to tol_scalar."""
    from speech_amd import ops
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    cfg = O.CONFIG_PAPER
    sd = cases.e2e_state(cfg)
    for k in sd:
        if k.endswith("nonlinear1d.weight") or k.endswith("prelu.weight"):
            sd[k].fill_(1.0)
    mixture, sources = cases.e2e_batch(cfg, 2, 16000, 7)
    res = {}
    default = ops.get_gemm_split()
    for mode in (False, True):
        ops.set_gemm_split(pieces if mode else 0)
        try:
            model = load_state(build(cfg), sd)
            out = model(mixture.to(dev()))
            loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev()))
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (out.detach(), loss.detach(), perm, {k: p.grad.clone() for k, p in model.named_parameters()})
        finally:
            ops.set_gemm_split(default)
    assert torch.equal(res[True][2], res[False][2])
    assert_close(res[True][0], res[False][0], 1e-4, "waveforms")
    assert_close(res[True][1], res[False][1], 1e-4, "loss")
    for k, g32 in res[False][3].items():
        assert_close(res[True][3][k], g32, tol_scalar if g32.numel() == 1 else tol_vec, "grad " + k)


@pytest.mark.parametrize("pieces", [2, 3])
def test_split_mode_training_steps_track_f32_mode(pieces):
    """Ten optimizer steps in each mode from the same weights and batches: the loss trajectories stay within 1e-3."""
    from speech_amd import ops
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.trainer import FlatTrainer
    cfg = O.CONFIG_SMALL
    traj = {}
    default = ops.get_gemm_split()
    for mode in (False, True):
        ops.set_gemm_split(pieces if mode else 0)
        try:
            model = load_state(build(cfg), cases.e2e_state(cfg))
            tr = FlatTrainer(model, PIT1d(NegSISDR(), 2))
            losses = []
            for i in range(10):
                mixture, sources = cases.e2e_batch(cfg, 4, 8000, 100 + i)
                loss, _ = tr.step(mixture.to(dev()), sources.to(dev()))
                losses.append(loss.item())
        finally:
            ops.set_gemm_split(default)
        traj[mode] = torch.tensor(losses)
    assert_close(traj[True], traj[False], 1e-3, "loss trajectory")
