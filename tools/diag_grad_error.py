"""Diagnostic: error of the GPU gradients and of the reference's fp32 gradients against a float64 oracle run."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cases, convtasnet_oracle as O
from conftest import rel_err
import test_parity_gpu as tp
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR

name, cfg = "e2e_small", O.CONFIG_SMALL
g = np.load(os.path.join(ROOT, "tests/golden", name + ".npz"))
sd = cases.e2e_state(cfg)
mix, src = cases.e2e_batch(cfg, int(g["batch"]), int(g["T"]), 0)
_, _, out64, g64 = O.loss_and_grads(mix.double(), src.double(), {k: v.double() for k, v in sd.items()}, cfg)
model = tp.load_state(tp.build(cfg), sd)
out = model(mix.cuda())
loss, perm = PIT1d(NegSISDR(), 2)(out, src.cuda())
loss.backward()
rows = []
for k, p in model.named_parameters():
    rows.append((rel_err(p.grad, g64[k]), rel_err(g["grad." + k], g64[k]), k))
rows.sort(reverse=True)
print("out: gpu %.2e ref %.2e" % (rel_err(out[..., ::8], out64[..., ::8]), rel_err(g["out"], out64[..., ::8])))
print("%-10s %-10s %s" % ("gpu/fp64", "ref/fp64", "param"))
for a, b, k in rows[:25]:
    print("%.3e  %.3e  %s" % (a, b, k))
