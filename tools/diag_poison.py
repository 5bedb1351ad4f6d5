"""Fills the caching allocator's free memory with NaN before one paper-size forward/backward: any tensor the
kernels read without having written it shows up as NaN in the outputs or gradients."""
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
which = sys.argv[1] if len(sys.argv) > 1 else "paper"
name, cfg = ("e2e_paper", O.CONFIG_PAPER) if which == "paper" else ("e2e_paper_causal3", O.CONFIG_CAUSAL3)
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
batch, T = int(g["batch"]), int(g["T"])
model = load_state(build(cfg), cases.e2e_state(cfg))
mixture, sources = cases.e2e_batch(cfg, batch, T, 0)
poison = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(12)]   # 12 GiB of NaN
torch.cuda.synchronize(); del poison
out = model(mixture.to(dev))
print("out nan:", bool(torch.isnan(out).any()))
loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev))
loss.backward(); torch.cuda.synchronize()
print("loss", float(loss.detach()), "golden", float(g["loss"]))
bad = [k for k, p in model.named_parameters() if torch.isnan(p.grad).any()]
print("params with NaN grad:", len(bad), bad[:12])
