"""Diagnostic: paper-size fixtures: strided gradient samples of the GPU run and of the reference's fp32 run (golden)
against a float64 oracle run.  usage: diag_paper_samples.py e2e_paper|e2e_paper_causal3"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import cases, convtasnet_oracle as O
import test_parity_gpu as tp
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
name = sys.argv[1]
cfg = O.CONFIG_PAPER if name == "e2e_paper" else O.CONFIG_CAUSAL3
g = np.load(os.path.join(ROOT, "tests/golden", name + ".npz"))
sd = cases.e2e_state(cfg)
mix, src = cases.e2e_batch(cfg, int(g["batch"]), int(g["T"]), 0)
_, _, out64, g64 = O.loss_and_grads(mix.double(), src.double(), {k: v.double() for k, v in sd.items()}, cfg)
model = tp.load_state(tp.build(cfg), sd)
out = model(mix.cuda())
loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, src.cuda())
loss.backward()
rows = []
for k, p in model.named_parameters():
    if p.numel() == 1:
        continue
    c64 = cases.checksum(g64[k])[2:]
    cg = cases.checksum(p.grad)[2:]
    cr = g["gradsum." + k][2:]
    sc = np.abs(c64).max()
    full = ((p.grad.cpu().double() - g64[k]).abs().max() / g64[k].abs().max()).item()
    rows.append((np.abs(cg - c64).max() / sc, np.abs(cr - c64).max() / sc, full, k))
rows.sort(reverse=True)
print("samples gpu/fp64  samples ref32/fp64  full-tensor gpu/fp64 (max-norm)")
for a, b, c, k in rows[:12]:
    print("%.3e  %.3e  %.3e  %s" % (a, b, c, k))
rows.sort(key=lambda r: -r[1])
print("worst reference rows")
for a, b, c, k in rows[:6]:
    print("%.3e  %.3e  %.3e  %s" % (a, b, c, k))
