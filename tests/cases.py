"""Inputs of the golden cases, regenerated from the seeds tools/gen_golden.py used."""
import numpy as np
import torch

from speech_amd.utils import synth

RESBLOCK_CASES = [
    # name, dilation, causal, dual_head
    ("d1_nc_dual", 1, False, True),
    ("d128_nc_dual", 128, False, True),
    ("d4_nc_single", 4, False, False),
    ("d1_c_dual", 1, True, True),
    ("d128_c_dual", 128, True, True),
    ("d2_c_single", 2, True, False),
]
RESBLOCK_DIMS = dict(Bc=16, H=32, Sc=24, T=301, b=2)


def norm_inputs():
    x = synth.normal((2, 8, 50), 11) * 0.7 + 3.0
    x[1] = 0.0
    xc = x.clone()
    xc[1] = synth.normal((8, 50), 12)
    gy = synth.normal((2, 8, 50), 13)
    x4 = synth.normal((2, 8, 5, 10), 14) + 0.5
    return dict(gln=x, cln=xc, gy=gy, x4=x4)


def norm_params(kind, C=8):
    """Parameters as fill_parameters(salt=3) gives them to a bare norm module."""
    if kind == "gln":
        sd = {"norm.weight": torch.zeros(C), "norm.bias": torch.zeros(C)}
    else:
        sd = {"gamma": torch.zeros(1, C, 1), "beta": torch.zeros(1, C, 1)}
    return synth.fill_parameters(sd, salt=3)


def resblock_state(causal, dual, Bc=16, H=32, Sc=24, P=3):
    sd = {}
    sd["bottleneck_conv1d.weight"] = torch.zeros(H, Bc, 1)
    sd["bottleneck_conv1d.bias"] = torch.zeros(H)
    sd["nonlinear1d.weight"] = torch.zeros(1)

    def norm(prefix):
        if causal:
            sd[prefix + "gamma"] = torch.zeros(1, H, 1)
            sd[prefix + "beta"] = torch.zeros(1, H, 1)
        else:
            sd[prefix + "norm.weight"] = torch.zeros(H)
            sd[prefix + "norm.bias"] = torch.zeros(H)
    norm("norm1d.")
    s = "separable_conv1d."
    sd[s + "depthwise_conv1d.weight"] = torch.zeros(H, 1, P)
    sd[s + "depthwise_conv1d.bias"] = torch.zeros(H)
    sd[s + "nonlinear1d.weight"] = torch.zeros(1)
    norm(s + "norm1d.")
    if dual:
        sd[s + "output_pointwise_conv1d.weight"] = torch.zeros(Bc, H, 1)
        sd[s + "output_pointwise_conv1d.bias"] = torch.zeros(Bc)
    sd[s + "skip_pointwise_conv1d.weight"] = torch.zeros(Sc, H, 1)
    sd[s + "skip_pointwise_conv1d.bias"] = torch.zeros(Sc)
    return synth.fill_parameters(sd, salt=5)


def resblock_inputs(dual):
    d = RESBLOCK_DIMS
    x = synth.normal((d["b"], d["Bc"], d["T"]), 21)
    g_skip = synth.normal((d["b"], d["Sc"], d["T"]), 22)
    g_out = synth.normal((d["b"], d["Bc"], d["T"]), 23) if dual else None
    return x, g_skip, g_out


def e2e_state(cfg):
    import convtasnet_oracle as O
    return synth.fill_parameters(O.make_state(cfg), salt=1)


def e2e_batch(cfg, batch, T, step=0):
    return synth.make_batch(batch, cfg["n_sources"], T, seed=1000 + step)


def checksum(t):
    f = torch.as_tensor(t).detach().double().flatten().cpu()
    n = f.numel()
    idx = torch.linspace(0, n - 1, 16).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])
