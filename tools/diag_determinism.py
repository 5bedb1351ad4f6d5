"""Runs the paper-size forward/backward several times in one process and prints, per run, the worst gradient
checksum deviation from the golden fixture and whether the gradients equal those of the first run bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cases
import convtasnet_oracle as O
from test_parity_gpu import build, load_state
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
dev = torch.device("cuda:0")
name, cfg = "e2e_paper", O.CONFIG_PAPER
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
batch, T = int(g["batch"]), int(g["T"])
first = None
keep = []
for run in range(5):
    model = load_state(build(cfg), cases.e2e_state(cfg))
    mixture, sources = cases.e2e_batch(cfg, batch, T, 0)
    out = model(mixture.to(dev))
    loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev))
    loss.backward(); torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    worst = max((abs(cases.checksum(v)[1] - g["gradsum." + k][1]) / g["gradsum." + k][1], k) for k, v in grads.items() if v.numel() > 1)
    line = "run %d worst abs-sum dev %.1e (%s)" % (run, worst[0], worst[1])
    if first is None:
        first = grads
    else:
        rel = max((float((grads[k] - first[k]).abs().max() / (first[k].abs().max() + 1e-30)), k) for k in grads)
        line += "  max rel diff to run 0: %.1e (%s)" % rel
    print(line, flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "keep":
        keep.append((model, out, loss))
