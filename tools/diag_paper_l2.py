"""Paper-width network, every gradient tensor: relative L2 error against a float64 oracle run, of this build (GPU) and of
the fp32 CPU oracle.  usage: [paper|causal3] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cases
import convtasnet_oracle as O
from conftest import rel_l2, rel_err
from test_parity_gpu import build, load_state
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
name = sys.argv[1] if len(sys.argv) > 1 else "paper"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
cfg = {"paper": O.CONFIG_PAPER, "causal3": O.CONFIG_CAUSAL3}[name]
dev = torch.device("cuda:0")
sd = cases.e2e_state(cfg)
mixture, sources = cases.e2e_batch(cfg, 1, T, 11)
_, _, out64, g64 = O.loss_and_grads(mixture.double(), sources.double(), {k: v.double() for k, v in sd.items()}, cfg)
_, _, _, g32 = O.loss_and_grads(mixture, sources, sd, cfg)
rows = []
for rep in range(2):
    model = load_state(build(cfg), sd)
    out = model(mixture.to(dev))
    loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev))
    loss.backward(); torch.cuda.synchronize()
    rows.append({k: (rel_l2(p.grad, g64[k]), rel_err(p.grad, g64[k])) for k, p in model.named_parameters() if p.numel() > 1})
tab = []
for k in rows[0]:
    tab.append((rows[0][k][0], rows[1][k][0], rel_l2(g32[k], g64[k]), rows[0][k][1], rel_err(g32[k], g64[k]), k))
tab.sort(reverse=True)
print("%-60s %9s %9s %9s | %9s %9s" % ("tensor", "gpu L2", "gpu L2 #2", "cpu32 L2", "gpu max", "cpu32 max"))
for a, a2, b, c, d, k in tab[:25]:
    print("%-60s %9.2e %9.2e %9.2e | %9.2e %9.2e" % (k[-60:], a, a2, b, c, d))
import statistics
print("median gpu L2 %.2e, median cpu32 L2 %.2e" % (statistics.median(t[0] for t in tab), statistics.median(t[2] for t in tab)))
