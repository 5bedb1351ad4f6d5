"""The filterbank variants on the HIP encoder / overlap-add kernels (speech_amd.models.filterbank: Fourier encoder and
decoder, pseudo-inverse decoders, gated encoder) and ConvTasNet built with them, against the reference's own outputs
(tests/golden/filterbank.npz): outputs and loss within 1e-4, permutations exact, gradients within 1e-4 for the
stand-alone modules and 5e-4 in relative L2 / 5e-3 in max-norm end to end (activation kinks, see assert_grad_close)."""
import numpy as np
import pytest
import torch

import filterbank_cases as FC
from conftest import assert_close, assert_grad_close
from speech_amd.utils import synth
from test_filterbank_cpu import as_real, probe_backward

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def load(module, g, prefix):
    sd = {k[len(prefix) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix + "_p_")}
    assert list(module.state_dict().keys()) == list(sd.keys())          # same names, same order as the reference
    for k, v in module.state_dict().items():
        assert v.shape == sd[k].shape and v.dtype == sd[k].dtype, k
    module.load_state_dict(sd)
    return module.to(dev())


def check(g, key, module, out, inputs, tol=1e-4):
    probe_backward(out)
    assert_close(as_real(out), g[key + "_out"], tol, key)
    for i, x in enumerate(inputs):
        assert_close(as_real(x.grad), g[key + "_dx{}".format(i)], tol, key + " dx")
    grads = dict(module.named_parameters())
    n = 0
    for k in g.files:
        if k.startswith(key + "_g_"):
            name = k[len(key) + 3:]
            assert_close(grads[name].grad, g[k], tol, key + " grad " + name)
            n += 1
    return n


@pytest.mark.parametrize("key", list(FC.FOURIER_ENCODERS))
def test_fourier_encoder(golden, key):
    from speech_amd.models.filterbank import FourierEncoder
    g = golden("filterbank")
    kw = FC.FOURIER_ENCODERS[key]
    m = load(FourierEncoder(FC.N_DFT, FC.L, stride=FC.S, **kw), g, "fenc_" + key)
    x = torch.from_numpy(g["x"]).to(dev()).requires_grad_(True)
    out = m(x)
    assert out.is_complex() == kw["return_complex"]
    assert check(g, "fenc_" + key, m, out, [x]) >= 1
    assert_close(m.get_basis().shape[0], 2 * (FC.N_DFT // 2 + 1 if kw["onesided"] else FC.N_DFT), 0, "basis rows")


@pytest.mark.parametrize("key", list(FC.FOURIER_DECODERS))
def test_fourier_decoder(golden, key):
    from speech_amd.models.filterbank import FourierDecoder
    g = golden("filterbank")
    kw, complex_in = FC.FOURIER_DECODERS[key]
    m = load(FourierDecoder(FC.N_DFT, FC.L, stride=FC.S, **kw), g, "fdec_" + key)
    z = FC.spectrum(synth, kw["onesided"], complex_in).to(dev()).requires_grad_(True)
    assert check(g, "fdec_" + key, m, m(z), [z]) >= 2


def test_analysis_synthesis_round_trip():
    """Fourier encoder followed by the Fourier decoder (optimal window) returns the interior of the signal."""
    from speech_amd.models.filterbank import FourierEncoder, FourierDecoder
    x = synth.normal((2, 1, 4000), 3, std=0.1).to(dev())
    for onesided in (True, False):
        enc = FourierEncoder(16, 16, stride=8, onesided=onesided, return_complex=False).to(dev())
        dec = FourierDecoder(16, 16, stride=8, onesided=onesided).to(dev())
        y = dec(enc(x))
        assert y.shape == x.shape
        assert_close(y[..., 16:-16], x[..., 16:-16], 1e-5, "round trip")


def test_pinv_decoders(golden):
    from speech_amd.models.filterbank import Encoder, FourierEncoder, GatedEncoder, PinvDecoder
    g = golden("filterbank")
    m = load(PinvDecoder(Encoder(1, FC.N_PINV, kernel_size=FC.L, stride=FC.S)), g, "pinv_trainable")
    z = synth.normal((2, FC.N_PINV, FC.FRAMES), 311, std=0.5).to(dev()).requires_grad_(True)
    out = m(z)
    probe_backward(out)
    assert_close(out, g["pinv_trainable_out"], 1e-4, "pinv")
    assert_close(z.grad, g["pinv_trainable_dx0"], 1e-4, "pinv dz")
    assert_close(m.weight.grad, g["pinv_trainable_g_weight"], 1e-4, "pinv dW")
    assert_close(m.get_basis(), g["pinv_trainable_basis"], 1e-4, "pinv basis")
    # pseudo-inverse synthesis of the encoder's own analysis returns the interior of the signal
    x = torch.from_numpy(g["x"]).to(dev())
    assert_close(m(m.encoder(x))[..., FC.L:-FC.L], x[..., FC.L:-FC.L], 1e-4, "pinv round trip")

    fenc = FourierEncoder(FC.N_DFT, FC.L, stride=FC.S, trainable=True, trainable_phase=True, onesided=False, return_complex=False)
    m = load(PinvDecoder(fenc), g, "pinv_fourier")
    z = FC.spectrum(synth, False, False).to(dev()).requires_grad_(True)
    check(g, "pinv_fourier", m, m(z), [z])

    with pytest.raises(ValueError):
        PinvDecoder(Encoder(1, FC.N_PINV, kernel_size=FC.L, stride=FC.S, nonlinear="relu"))
    with pytest.raises(ValueError):
        PinvDecoder(Encoder(1, 8, kernel_size=FC.L, stride=FC.S))
    with pytest.raises(ValueError):
        PinvDecoder(FourierEncoder(FC.N_DFT, FC.L, stride=FC.S))
    with pytest.raises(TypeError):
        PinvDecoder(GatedEncoder(1, FC.N_PINV, kernel_size=FC.L, stride=FC.S))


def test_gated_encoder(golden):
    from speech_amd.models.filterbank import GatedEncoder
    g = golden("filterbank")
    m = load(GatedEncoder(1, FC.N_PINV, kernel_size=FC.L, stride=FC.S), g, "gated")
    x = torch.from_numpy(g["x"]).to(dev()).requires_grad_(True)
    assert check(g, "gated", m, m(x), [x]) == 2


@pytest.mark.parametrize("key", list(FC.E2E) + list(FC.E2E_NONSEPARABLE))
def test_conv_tasnet_with_filterbank_variants(golden, key):
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.models.conv_tasnet import ConvTasNet
    g = golden("nonseparable" if key in FC.E2E_NONSEPARABLE else "filterbank")
    cfg = dict(FC.E2E_BASE, **(FC.E2E_NONSEPARABLE[key] if key in FC.E2E_NONSEPARABLE else FC.E2E[key]))
    model = ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis=cfg["enc_basis"],
                       dec_basis=cfg["dec_basis"], enc_nonlinear=cfg.get("enc_nonlinear"),
                       in_channels=cfg.get("in_channels") or 1, window_fn=cfg.get("window_fn"),
                       enc_onesided=cfg.get("enc_onesided"), enc_return_complex=cfg.get("enc_return_complex"),
                       sep_hidden_channels=cfg["sep_hidden_channels"], sep_bottleneck_channels=cfg["sep_bottleneck_channels"],
                       sep_skip_channels=cfg["sep_skip_channels"], sep_kernel_size=cfg["sep_kernel_size"],
                       sep_num_blocks=cfg["sep_num_blocks"], sep_num_layers=cfg["sep_num_layers"], dilated=cfg.get("dilated", True),
                       separable=cfg.get("separable", True), sep_nonlinear="prelu", sep_norm=True,
                       mask_nonlinear=cfg["mask_nonlinear"], causal=cfg["causal"], n_sources=cfg["n_sources"])
    assert list(model.state_dict().keys()) == [str(k) for k in g["e2e_" + key + "_keys"]]
    load(model, g, "e2e_" + key)
    mixture, sources = FC.e2e_batch(synth, cfg)
    out = model(mixture.to(dev()))
    loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev()))
    loss.backward()
    assert np.array_equal(perm.cpu().numpy(), g["e2e_" + key + "_perm"])
    assert_close(out, g["e2e_" + key + "_out"], 1e-4, "waveforms")
    assert_close(loss, g["e2e_" + key + "_loss"], 1e-4, "loss")
    grads = dict(model.named_parameters())
    n = 0
    for k in g.files:
        if k.startswith("e2e_" + key + "_g_"):
            name = k[len("e2e_" + key) + 3:]
            assert_grad_close(grads[name].grad, g[k], "grad " + name)
            n += 1
    assert n > 20
    # forward() may take the fused schedule, extract_latent() always goes module by module: same answer
    out2, latent = model.extract_latent(mixture.to(dev()))
    assert_close(out2, out.detach(), 1e-5, "fused vs modular")
    assert latent.shape[:3] == (2, cfg["n_sources"], cfg["n_basis"])
    cfg_out = model.get_config()
    assert cfg_out["enc_basis"] == cfg["enc_basis"] and cfg_out["dec_basis"] == cfg["dec_basis"]
