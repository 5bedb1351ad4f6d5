"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference); the fixtures it
writes are plain arrays (inputs are regenerated from seeds by
speech_amd.utils.synth, only the reference's outputs are stored).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
sys.path.insert(1, ROOT)
# utils.audio imports torchaudio at module top for an evaluation helper that is
# never called here; the package is not installed, so register an empty module.
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.conv_tasnet import ConvTasNet  # noqa: E402  (reference)
from models.tdcn import ResidualBlock1d  # noqa: E402  (reference)
from modules.norm import GlobalLayerNorm, CumulativeLayerNorm1d  # noqa: E402  (reference)
from criterion.sdr import NegSISDR, SISDR, sisdr  # noqa: E402  (reference)
from criterion.pit import PIT1d  # noqa: E402  (reference)

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "speech_amd", "utils", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
META = np.array("reference run on torch {} CPU fp32 (reference pins torch 1.10)".format(torch.__version__))


def np_(t):
    return t.detach().cpu().numpy().copy()


def checksum(t):
    """sum, abs-sum and 16 strided samples of a tensor."""
    f = t.detach().double().flatten()
    n = f.numel()
    idx = torch.linspace(0, n - 1, 16).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


# ----------------------------------------------------------------- norms
def gen_norm():
    d = {"meta": META}
    x = synth.normal((2, 8, 50), 11) * 0.7 + 3.0          # non-zero mean
    x[1] = 0.0                                             # an all-zero sample
    for name, mod in (("gln", GlobalLayerNorm(8)), ("cln", CumulativeLayerNorm1d(8))):
        synth.fill_parameters(mod, salt=3)
        xi = x.clone().requires_grad_(True)
        if name == "cln":
            xi = (x.clone() + 0.0)
            xi[1] = synth.normal((8, 50), 12)              # cLN of an all-zero sample is 0/eps; keep it finite
            xi.requires_grad_(True)
        y = mod(xi)
        gy = synth.normal(tuple(y.shape), 13)
        y.backward(gy)
        d[name + "_y"] = np_(y)
        d[name + "_dx"] = np_(xi.grad)
        for k, p in mod.named_parameters():
            d["{}_d_{}".format(name, k)] = np_(p.grad)
    # all-zero sample through gLN forward only (output = bias)
    g = GlobalLayerNorm(8)
    synth.fill_parameters(g, salt=3)
    d["gln_zero_y"] = np_(g(torch.zeros(1, 8, 50)))
    # 4-D cLN input (B, C, S, chunk): norm.py:71-74
    c = CumulativeLayerNorm1d(8)
    synth.fill_parameters(c, salt=3)
    x4 = synth.normal((2, 8, 5, 10), 14) + 0.5
    d["cln4_y"] = np_(c(x4))
    np.savez(os.path.join(OUT, "norm.npz"), **d)


# ----------------------------------------------------------------- residual block
RESBLOCK_CASES = [
    # name, dilation, causal, dual_head
    ("d1_nc_dual", 1, False, True),
    ("d128_nc_dual", 128, False, True),
    ("d4_nc_single", 4, False, False),
    ("d1_c_dual", 1, True, True),
    ("d128_c_dual", 128, True, True),
    ("d2_c_single", 2, True, False),
]


def gen_resblock():
    d = {"meta": META}
    Bc, H, Sc, T, b = 16, 32, 24, 301, 2
    for name, dil, causal, dual in RESBLOCK_CASES:
        m = ResidualBlock1d(Bc, hidden_channels=H, skip_channels=Sc, kernel_size=3, stride=1, dilation=dil,
                            separable=True, causal=causal, nonlinear="prelu", norm=True, dual_head=dual)
        synth.fill_parameters(m, salt=5)
        x = synth.normal((b, Bc, T), 21).requires_grad_(True)
        out, skip = m(x)
        g_skip = synth.normal(tuple(skip.shape), 22)
        loss = (skip * g_skip).sum()
        if out is not None:
            g_out = synth.normal(tuple(out.shape), 23)
            loss = loss + (out * g_out).sum()
            d[name + "_out"] = np_(out)
        loss.backward()
        d[name + "_skip"] = np_(skip)
        d[name + "_dx"] = np_(x.grad)
        for k, p in m.named_parameters():
            d["{}_d_{}".format(name, k)] = np_(p.grad)
    np.savez(os.path.join(OUT, "resblock.npz"), **d)


# ----------------------------------------------------------------- PIT SI-SDR
def gen_pit():
    d = {"meta": META}
    T = 4000
    cases = {}
    for S in (2, 3):
        tgt = synth.normal((4, S, T), 31 + S, std=0.1)
        noise = synth.normal((4, S, T), 41 + S, std=0.03)
        perm = [list(np.roll(np.arange(S), k % S)) for k in range(4)]
        est = torch.stack([tgt[i, perm[i]] for i in range(4)]) * 0.8 + noise
        cases["rand{}".format(S)] = (est, tgt)
    tgt = synth.normal((2, 2, T), 51, std=0.1)
    tgt[:, 1] = tgt[:, 0]                                  # identical targets -> exact tie -> perm [0, 1]
    cases["tie"] = (synth.normal((2, 2, T), 52, std=0.1), tgt)
    tgt = synth.normal((2, 2, T), 53, std=0.1)
    cases["swap"] = (tgt[:, [1, 0]] + synth.normal((2, 2, T), 54, std=0.01), tgt)
    tgt = synth.normal((2, 2, T), 55, std=0.1)
    tgt[0, 1] = 0.0                                        # a silent target
    cases["silent"] = (synth.normal((2, 2, T), 56, std=0.1), tgt)
    for name, (est, tgt) in cases.items():
        S = est.shape[1]
        est = est.clone().requires_grad_(True)
        crit = PIT1d(NegSISDR(), S)
        loss_b, perm_b = crit(est, tgt, batch_mean=False)
        loss, perm = crit(est, tgt)
        loss.backward()
        d[name + "_est"] = np_(est)
        d[name + "_tgt"] = np_(tgt)
        d[name + "_loss_b"] = np_(loss_b)
        d[name + "_loss"] = np_(loss)
        d[name + "_perm"] = np_(perm)
        d[name + "_dest"] = np_(est.grad)
        d[name + "_pair"] = np_(torch.stack([sisdr(est.detach(), tgt[:, [j] * S]) for j in range(S)], dim=-1))
    e, t = cases["rand2"]
    d["sisdr_rand2"] = np_(SISDR()(e, t, batch_mean=False))
    np.savez(os.path.join(OUT, "pit.npz"), **d)


# ----------------------------------------------------------------- end to end
def build(cfg):
    m = ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis="trainable",
                   dec_basis="trainable", enc_nonlinear=cfg["enc_nonlinear"],
                   sep_hidden_channels=cfg["sep_hidden_channels"],
                   sep_bottleneck_channels=cfg["sep_bottleneck_channels"],
                   sep_skip_channels=cfg["sep_skip_channels"], sep_kernel_size=cfg["sep_kernel_size"],
                   sep_num_blocks=cfg["sep_num_blocks"], sep_num_layers=cfg["sep_num_layers"], dilated=True,
                   separable=True, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear=cfg["mask_nonlinear"],
                   causal=cfg["causal"], n_sources=cfg["n_sources"])
    synth.fill_parameters(m, salt=1)
    return m


def gen_e2e(name, cfg, batch, T, steps, full):
    torch.manual_seed(0)
    m = build(cfg)
    crit = PIT1d(NegSISDR(), cfg["n_sources"])
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    d = {"meta": META, "batch": batch, "T": T}
    for step in range(steps):
        mixture, sources = synth.make_batch(batch, cfg["n_sources"], T, seed=1000 + step)
        out = m(mixture)
        loss, perm = crit(out, sources)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            d["loss"] = np_(loss)
            d["perm"] = np_(perm)
            d["out_stride"] = 8
            d["out"] = np_(out)[..., ::8].copy()
            d["out_sum"] = checksum(out)
            for k, p in m.named_parameters():
                if full:
                    d["grad." + k] = np_(p.grad)
                else:
                    d["gradsum." + k] = checksum(p.grad)
            d["grad_norm"] = np.array(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item())
        else:
            d["loss_step{}".format(step)] = np_(loss)
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        if full or step == steps - 1:
            for k, p in m.named_parameters():
                key = "param_step{}.{}".format(step + 1, k)
                d[key] = np_(p) if full else checksum(p)
    np.savez(os.path.join(OUT, name + ".npz"), **d)
    print(name, "loss", d["loss"], "perm", d["perm"].tolist())


def main():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import convtasnet_oracle as O
    if "--causal3-full" in sys.argv:      # only the full-length causal fixture (the others are unchanged)
        gen_e2e("e2e_paper_causal3_full", O.CONFIG_CAUSAL3, batch=1, T=32000, steps=1, full=False)
        return
    gen_norm()
    gen_resblock()
    gen_pit()
    gen_e2e("e2e_small", O.CONFIG_SMALL, batch=2, T=32000, steps=2, full=True)
    small_c3 = dict(O.CONFIG_SMALL, causal=True, n_sources=3, enc_nonlinear="relu")
    gen_e2e("e2e_small_causal3", small_c3, batch=2, T=8003, steps=2, full=True)
    gen_e2e("e2e_paper", O.CONFIG_PAPER, batch=1, T=32000, steps=1, full=False)
    gen_e2e("e2e_paper_causal3", O.CONFIG_CAUSAL3, batch=1, T=16000, steps=1, full=False)
    gen_e2e("e2e_paper_causal3_full", O.CONFIG_CAUSAL3, batch=1, T=32000, steps=1, full=False)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
