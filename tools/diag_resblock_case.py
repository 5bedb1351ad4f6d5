"""One residual layer against the float64 oracle: per-tensor errors, with the PReLU slopes as filled and at 1 (kink-free).
usage: b Bc H Sc T dilation causal dual"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases
import convtasnet_oracle as O
from speech_amd.models.tdcn import ResidualBlock1d
from speech_amd.utils import synth
b, Bc, H, Sc, T, dil = (int(v) for v in sys.argv[1:7])
causal, dual = sys.argv[7] == "1", sys.argv[8] == "1"
dev = torch.device("cuda:0")


def rel(a, r):
    r = r.double().cpu(); a = a.double().cpu()
    return ((a - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()


for smooth in (False, True):
    sd = cases.resblock_state(causal, dual, Bc=Bc, H=H, Sc=Sc)
    if smooth:
        for k in sd:
            if k.endswith("nonlinear1d.weight"):
                sd[k].fill_(1.0)
    x = synth.normal((b, Bc, T), 91)
    g_skip, g_out = synth.normal((b, Sc, T), 92), synth.normal((b, Bc, T), 93)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    out64, skip64 = O.residual_block(x64, sd64, "", dil, causal, dual)
    ((skip64 * g_skip.double()).sum() + ((out64 * g_out.double()).sum() if dual else 0.0)).backward()
    mod = ResidualBlock1d(Bc, hidden_channels=H, skip_channels=Sc, kernel_size=3, stride=1, dilation=dil, separable=True,
                          causal=causal, nonlinear="prelu", norm=True, dual_head=dual)
    mod.load_state_dict(sd); mod.to(dev)
    xg = x.to(dev).requires_grad_(True)
    out, skip = mod(xg)
    loss = (skip * g_skip.to(dev)).sum()
    if dual:
        loss = loss + (out * g_out.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    print("smooth" if smooth else "kinked", "skip %.2e" % rel(skip, skip64), "out %.2e" % (rel(out, out64) if dual else 0),
          "dx %.2e" % rel(xg.grad, x64.grad))
    e = (xg.grad.double().cpu() - x64.grad).abs()
    idx = torch.nonzero(e > 1e-3 * x64.grad.abs().max())
    print("  dx elements off by > 1e-3 of max:", idx.shape[0], idx[:6].tolist())
    print("  " + " ".join("%s %.1e" % (k.split(".")[-2][:6] + "." + k.split(".")[-1][:4], rel(p.grad, sd64[k].grad))
                          for k, p in mod.named_parameters()))
