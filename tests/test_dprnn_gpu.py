"""DPRNN-TasNet (speech_amd/models/dprnn_tasnet.py) and the chunking kernels against outputs of the reference's own
modules (tests/golden/dprnn.npz from tools/gen_golden_dprnn.py): Segment1d / OverlapAdd1d with the separator's
paddings, forward and backward, bit-exact; whole networks (small bidirectional, small causal 3-speaker, the recipe's
default configuration on a 4 s 16 kHz mixture) -- waveforms and loss at 1e-4, permutations bit-exact, gradients at 1e-3
of each tensor's largest entry (the recurrences run on the persistent LSTM kernels, tests/test_lstm_gpu.py, against PyTorch's
CPU LSTM in the fixture)."""
import numpy as np
import pytest
import torch

import dprnn_cases as cases
from conftest import assert_close
from speech_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", list(cases.SEGMENT_CASES))
def test_segment_and_overlap_add_match_reference(golden, name):
    from speech_amd.functional import SegmentFn, OverlapAddFn
    g = golden("dprnn")
    B, C, T, K, P = cases.SEGMENT_CASES[name]
    padding = (P - (T - K) % P) % P
    pl = padding // 2
    S = (T + padding - K) // P + 1
    x = synth.normal((B, C, T), 31).to(DEV).requires_grad_(True)
    y = SegmentFn.apply(x, K, P, S, pl)
    gy = synth.normal(tuple(y.shape), 32).to(DEV)
    y.backward(gy)
    assert tuple(y.shape) == g[name + "_y"].shape
    assert torch.equal(y.cpu(), torch.from_numpy(g[name + "_y"]))
    assert_close(x.grad, g[name + "_dx"], 1e-6, "segment backward")
    yy = gy.clone().requires_grad_(True)
    z = OverlapAddFn.apply(yy, P, T, pl)
    gz = synth.normal(tuple(z.shape), 33).to(DEV)
    z.backward(gz)
    assert_close(z, g[name + "_z"], 1e-6, "overlap-add")
    assert torch.equal(yy.grad.cpu(), torch.from_numpy(g[name + "_dy"]))


def test_transform_modules_without_padding():
    import torch.nn.functional as F
    from speech_amd.models.transform import Segment1d, OverlapAdd1d
    x = synth.normal((2, 3, 45), 5)
    ref = F.unfold(x.view(2, 3, 45, 1), kernel_size=(10, 1), stride=(5, 1)).view(2, 3, 10, -1).permute(0, 1, 3, 2)
    y = Segment1d(10, 5)(x.to(DEV))
    assert torch.equal(y.cpu(), ref.contiguous())
    back = OverlapAdd1d(10, 5)(y)
    ref_back = F.fold(ref.permute(0, 1, 3, 2).reshape(2, 30, -1), kernel_size=(10, 1), stride=(5, 1),
                      output_size=(45, 1)).squeeze(3)
    assert_close(back, ref_back, 1e-6, "overlap-add of the segmentation")


@pytest.mark.parametrize("name", list(cases.E2E))
def test_dprnn_tasnet_matches_reference(golden, name):
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.models.dprnn_tasnet import DPRNNTasNet
    g = golden("dprnn")
    cfg, B, T = cases.E2E[name]
    model = DPRNNTasNet(**cfg)
    assert list(model.state_dict().keys()) == g[name + "_keys"].tolist()
    assert model.num_parameters == int(g[name + "_nparams"])
    synth.fill_parameters(model, salt=9)
    model.to(DEV)
    S = cfg["n_sources"]
    src = (0.1 * synth.normal((B, S, T), 41)).to(DEV)
    mix = src.sum(1, keepdim=True)
    out, latent = model.extract_latent(mix)
    loss, perm = PIT1d(NegSISDR(), n_sources=S)(out, src)
    loss.backward()
    torch.cuda.synchronize()
    assert perm.cpu().tolist() == g[name + "_perm"].tolist()
    assert_close(loss.reshape(1), g[name + "_loss"].reshape(1), 1e-4, "loss")
    if name == "paper":
        assert_close(out[0, :, ::997], g[name + "_out_samples"], 1e-4, "waveform samples")
        cs = torch.stack([torch.stack([p.grad.double().sum(), p.grad.double().abs().sum()]) for p in model.parameters()])
        assert_close(cs[:, 1], g[name + "_grads"][:, 1], 1e-3, "gradient abs-sums")
    else:
        assert_close(out, g[name + "_out"], 1e-4, "waveforms")
        assert_close(latent, g[name + "_latent"], 1e-4, "latent")
        for k, p in model.named_parameters():
            assert_close(p.grad, g[name + "_grad_" + k], 1e-3, "grad " + k)
