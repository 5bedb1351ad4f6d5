"""Diagnostic: one residual layer at T=3999, GPU vs float64 vs torch fp32 CPU, including intermediates."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cases, convtasnet_oracle as O
from conftest import rel_err
from speech_amd.utils import synth
from speech_amd.models.tdcn import ResidualBlock1d

Bc, H, Sc, T, b = 32, 64, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 3999, 2
dil = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sd = cases.resblock_state(False, True, Bc, H, Sc)
x = synth.normal((b, Bc, T), 21)
gs = synth.normal((b, Sc, T), 22)
go = synth.normal((b, Bc, T), 23)

def run(dtype):
    p = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    xx = x.to(dtype).clone().requires_grad_(True)
    out, skip = O.residual_block(xx, p, "", dil, False, True)
    ((skip * gs.to(dtype)).sum() + (out * go.to(dtype)).sum()).backward()
    return out, skip, xx.grad, {k: v.grad for k, v in p.items()}

o64, s64, dx64, g64 = run(torch.float64)
o32, s32, dx32, g32 = run(torch.float32)
mod = ResidualBlock1d(Bc, hidden_channels=H, skip_channels=Sc, kernel_size=3, stride=1, dilation=dil, separable=True,
                      causal=False, nonlinear="prelu", norm=True, dual_head=True)
mod.load_state_dict(sd); mod.cuda()
xg = x.cuda().requires_grad_(True)
og, sg = mod(xg)
((sg * gs.cuda()).sum() + (og * go.cuda()).sum()).backward()
print("T=%d d=%d   %-10s %-10s" % (T, dil, "gpu/fp64", "cpu32/fp64"))
print("out   %.3e  %.3e" % (rel_err(og, o64), rel_err(o32, o64)))
print("skip  %.3e  %.3e" % (rel_err(sg, s64), rel_err(s32, s64)))
print("dx    %.3e  %.3e" % (rel_err(xg.grad, dx64), rel_err(dx32, dx64)))
for k, p in mod.named_parameters():
    print("%.3e  %.3e  %s" % (rel_err(p.grad, g64[k]), rel_err(g32[k], g64[k]), k))
