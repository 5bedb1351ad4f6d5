"""Generate tests/golden/nonseparable.npz by running the REFERENCE's ConvTasNet with separable=False (dense dilated
convolution heads in every residual layer, src/models/tdcn.py:102-105, :136-142) on CPU: forward, PIT SI-SDR loss and
every gradient, non-causal / 2 sources and causal / 3 sources.  Runs only in the build container (needs
/root/reference); the fixture holds plain arrays.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_nonseparable.py
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

OUT = os.path.join(ROOT, "tests", "golden", "nonseparable.npz")


def np_(t):
    return t.detach().cpu().numpy()


def main():
    d = {"meta": np.array("reference run on torch {} CPU fp32 (reference pins torch 1.10)".format(torch.__version__))}
    for key, extra in FC.E2E_NONSEPARABLE.items():
        torch.manual_seed(111)
        cfg = dict(FC.E2E_BASE, **extra)
        model = ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis=cfg["enc_basis"],
                           dec_basis=cfg["dec_basis"], enc_nonlinear=cfg.get("enc_nonlinear"),
                           sep_hidden_channels=cfg["sep_hidden_channels"],
                           sep_bottleneck_channels=cfg["sep_bottleneck_channels"],
                           sep_skip_channels=cfg["sep_skip_channels"], sep_kernel_size=cfg["sep_kernel_size"],
                           sep_num_blocks=cfg["sep_num_blocks"], sep_num_layers=cfg["sep_num_layers"], dilated=True,
                           separable=False, sep_nonlinear="prelu", sep_norm=True,
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
            d["e2e_" + key + "_g_" + name] = np_(p.grad)
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, {k: v.shape for k, v in d.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
