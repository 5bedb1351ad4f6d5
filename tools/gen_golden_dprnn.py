"""Generate tests/golden/dprnn.npz by running the REFERENCE's DPRNN-TasNet and chunking modules on CPU
(src/models/dprnn_tasnet.py, dprnn.py, transform.py).  Runs only in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_dprnn.py
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
sys.path.insert(1, ROOT)
sys.path.insert(2, os.path.join(ROOT, "tests"))
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from models.dprnn_tasnet import DPRNNTasNet  # noqa: E402  (reference)
from models.transform import Segment1d, OverlapAdd1d  # noqa: E402  (reference)
from criterion.sdr import NegSISDR  # noqa: E402  (reference)
from criterion.pit import PIT1d  # noqa: E402  (reference)

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "speech_amd", "utils", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
import dprnn_cases as cases  # noqa: E402

torch.set_num_threads(8)


def np_(t):
    return t.detach().cpu().numpy().copy()


def checksum(t):
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, 16).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def main():
    d = {"meta": np.array("reference run on torch {} CPU fp32".format(torch.__version__))}
    # ---- chunking: forward and backward of Segment1d / OverlapAdd1d, alone and with the separator's paddings
    for name, (B, C, T, K, P) in cases.SEGMENT_CASES.items():
        x = synth.normal((B, C, T), 31).requires_grad_(True)
        padding = (P - (T - K) % P) % P
        pl = padding // 2
        xp = F.pad(x, (pl, padding - pl))
        y = Segment1d(K, P)(xp)
        gy = synth.normal(tuple(y.shape), 32)
        y.backward(gy)
        d[name + "_y"], d[name + "_dx"] = np_(y), np_(x.grad)
        yy = gy.clone().requires_grad_(True)
        z = F.pad(OverlapAdd1d(K, P)(yy), (-pl, -(padding - pl)))
        gz = synth.normal(tuple(z.shape), 33)
        z.backward(gz)
        d[name + "_z"], d[name + "_dy"] = np_(z), np_(yy.grad)
    # ---- end to end
    for name, (cfg, B, T) in cases.E2E.items():
        model = DPRNNTasNet(**cfg)
        synth.fill_parameters(model, salt=9)
        S = cfg["n_sources"]
        src = 0.1 * synth.normal((B, S, T), 41)
        mix = src.sum(1, keepdim=True)
        out, latent = model.extract_latent(mix)
        loss, perm = PIT1d(NegSISDR(), n_sources=S)(out, src)
        loss.backward()
        d[name + "_loss"], d[name + "_perm"] = np_(loss), np_(perm)
        d[name + "_keys"] = np.array(list(model.state_dict().keys()))
        d[name + "_nparams"] = np.array(model.num_parameters)
        if name == "paper":
            d[name + "_out"] = checksum(out)
            d[name + "_out_samples"] = np_(out[0, :, ::997])
            d[name + "_grads"] = np.stack([checksum(p.grad) for p in model.parameters()])
        else:
            d[name + "_out"], d[name + "_latent"] = np_(out), np_(latent)
            for k, p in model.named_parameters():
                d[name + "_grad_" + k] = np_(p.grad)
        print(name, "loss", loss.item(), "perm", perm.tolist(), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dprnn.npz"), **d)
    print("wrote dprnn.npz with", len(d), "arrays")


if __name__ == "__main__":
    main()
