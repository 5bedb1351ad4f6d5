"""Diagnostic: paper config, scalar PReLU-slope gradients: GPU and reference-fp32 (golden) against float64."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, time
import cases, convtasnet_oracle as O
from conftest import rel_err, rel_l2
import test_parity_gpu as tp
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
cfg = O.CONFIG_PAPER
g = np.load(os.path.join(ROOT, "tests/golden/e2e_paper.npz"))
sd = cases.e2e_state(cfg)
mix, src = cases.e2e_batch(cfg, 1, 32000, 0)
t0 = time.time()
_, _, out64, g64 = O.loss_and_grads(mix.double(), src.double(), {k: v.double() for k, v in sd.items()}, cfg)
print("fp64 oracle %.1fs" % (time.time() - t0))
model = tp.load_state(tp.build(cfg), sd)
out = model(mix.cuda())
loss, perm = PIT1d(NegSISDR(), 2)(out, src.cuda())
loss.backward()
rows = []
for k, p in model.named_parameters():
    ref64 = g64[k]
    cs_ref = g["gradsum." + k]
    if p.numel() == 1:
        t = ref64.item()
        rows.append((abs(p.grad.item() - t) / abs(t), abs(cs_ref[0] - t) / abs(t), k))
rows.sort(reverse=True)
print("scalar grads: gpu/fp64  ref32/fp64")
for a, b, k in rows[:12]:
    print("%.3e  %.3e  %s" % (a, b, k))
worst = sorted(((rel_l2(p.grad, g64[k]), rel_err(p.grad, g64[k]), k) for k, p in model.named_parameters()), reverse=True)
print("all grads gpu/fp64: L2, max")
for a, b, k in worst[:8]:
    print("%.3e  %.3e  %s" % (a, b, k))
