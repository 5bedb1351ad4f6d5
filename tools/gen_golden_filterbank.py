"""Generate tests/golden/filterbank.npz by running the REFERENCE's own filterbank modules and ConvTasNet on CPU:
Fourier encoder / decoder (one- and two-sided, real and complex, trainable frequency and phase), pseudo-inverse
decoders, the gated encoder, and ConvTasNet forward + loss + gradients with non-default filterbank pairs, the non-dilated TCN, two microphones and a complex latent.
Runs only in the build container (needs /root/reference); the fixture holds plain arrays.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_filterbank.py
"""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
sys.path.insert(1, os.path.join(ROOT, "tests"))
# utils.audio imports torchaudio at module top for an evaluation helper that is never called here
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.filterbank import FourierEncoder, FourierDecoder, Encoder, PinvDecoder, GatedEncoder  # noqa: E402  (reference)
from models.conv_tasnet import ConvTasNet  # noqa: E402  (reference)
from criterion.sdr import NegSISDR  # noqa: E402  (reference)
from criterion.pit import PIT1d  # noqa: E402  (reference)

import filterbank_cases as FC  # noqa: E402  (tests/filterbank_cases.py: the case definitions shared with the tests)

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "speech_amd", "utils", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.path.join(ROOT, "tests", "golden", "filterbank.npz")


def np_(t):
    t = t.detach().cpu()
    if torch.is_complex(t):
        t = torch.view_as_real(t)
    return t.numpy().copy()


def record(d, key, module, out, inputs):
    """Output, gradients of sum(out * probe) w.r.t. every trainable parameter and the inputs, and the parameters."""
    real = torch.view_as_real(out) if torch.is_complex(out) else out
    probe = synth.normal(tuple(real.shape), 977, std=1.0)
    (real * probe).sum().backward()
    d[key + "_out"] = np_(out)
    for name, p in module.state_dict().items():
        d[key + "_p_" + name] = np_(p)
    for name, p in module.named_parameters():
        if p.requires_grad:
            d[key + "_g_" + name] = np_(p.grad)
    for i, x in enumerate(inputs):
        d[key + "_dx{}".format(i)] = np_(x.grad)


def main():
    d = {"meta": np.array("reference run on torch {} CPU fp32 (reference pins torch 1.10)".format(torch.__version__))}
    x = synth.normal((2, 1, FC.T), 301, std=0.1)
    d["x"] = np_(x)

    for key, kw in FC.FOURIER_ENCODERS.items():
        m = FourierEncoder(FC.N_DFT, FC.L, stride=FC.S, **kw)
        FC.perturb(m, synth)
        xi = x.clone().requires_grad_(True)
        record(d, "fenc_" + key, m, m(xi), [xi])

    for key, (kw, complex_in) in FC.FOURIER_DECODERS.items():
        m = FourierDecoder(FC.N_DFT, FC.L, stride=FC.S, **kw)
        FC.perturb(m, synth)
        z = FC.spectrum(synth, kw["onesided"], complex_in).requires_grad_(True)
        record(d, "fdec_" + key, m, m(z), [z])

    enc = Encoder(1, FC.N_PINV, kernel_size=FC.L, stride=FC.S)
    synth.fill_parameters(enc, salt=5)
    m = PinvDecoder(enc)
    z = synth.normal((2, FC.N_PINV, FC.FRAMES), 311, std=0.5).requires_grad_(True)
    record(d, "pinv_trainable", m, m(z), [z])
    d["pinv_trainable_basis"] = np_(m.get_basis())

    fenc = FourierEncoder(FC.N_DFT, FC.L, stride=FC.S, trainable=True, trainable_phase=True, onesided=False, return_complex=False)
    FC.perturb(fenc, synth)
    m = PinvDecoder(fenc)
    z = FC.spectrum(synth, False, False).requires_grad_(True)
    record(d, "pinv_fourier", m, m(z), [z])

    m = GatedEncoder(1, FC.N_PINV, kernel_size=FC.L, stride=FC.S)
    synth.fill_parameters(m, salt=6)
    xi = x.clone().requires_grad_(True)
    record(d, "gated", m, m(xi), [xi])

    # --- ConvTasNet with non-default filterbanks: forward, PIT SI-SDR loss, every gradient
    for key, extra in FC.E2E.items():
        torch.manual_seed(111)
        cfg = dict(FC.E2E_BASE, **extra)
        model = ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis=cfg["enc_basis"],
                           dec_basis=cfg["dec_basis"], enc_nonlinear=cfg.get("enc_nonlinear"),
                           in_channels=cfg.get("in_channels") or 1, window_fn=cfg.get("window_fn"), enc_onesided=cfg.get("enc_onesided"),
                           enc_return_complex=cfg.get("enc_return_complex"),
                           sep_hidden_channels=cfg["sep_hidden_channels"],
                           sep_bottleneck_channels=cfg["sep_bottleneck_channels"],
                           sep_skip_channels=cfg["sep_skip_channels"], sep_kernel_size=cfg["sep_kernel_size"],
                           sep_num_blocks=cfg["sep_num_blocks"], sep_num_layers=cfg["sep_num_layers"],
                           dilated=cfg.get("dilated", True), separable=True, sep_nonlinear="prelu", sep_norm=True,
                           mask_nonlinear=cfg["mask_nonlinear"], causal=cfg["causal"], n_sources=cfg["n_sources"])
        FC.fill_model(model, synth)
        mixture, sources = FC.e2e_batch(synth, cfg)
        out = model(mixture)
        loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources)
        loss.backward()
        d["e2e_" + key + "_out"] = np_(out)
        d["e2e_" + key + "_loss"] = np_(loss)
        d["e2e_" + key + "_perm"] = np_(perm)
        d["e2e_" + key + "_keys"] = np.array(list(model.state_dict().keys()))
        for name, p in model.state_dict().items():
            d["e2e_" + key + "_p_" + name] = np_(p)
        for name, p in model.named_parameters():
            if p.requires_grad:
                d["e2e_" + key + "_g_" + name] = np_(p.grad)

    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(d), "arrays")


if __name__ == "__main__":
    main()
