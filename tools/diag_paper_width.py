"""Paper-size golden comparison under both GEMM tile widths: prints the worst gradient checksum deviations
(abs-sum and sum, relative to the abs-sum) so that run-to-run and width-to-width spread can be told from a bug."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cases
import convtasnet_oracle as O
from speech_amd import _lib
from test_parity_gpu import build, load_state
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
dev = torch.device("cuda:0")
lib = _lib.load()
for name, cfg in [("e2e_paper", O.CONFIG_PAPER), ("e2e_paper_causal3", O.CONFIG_CAUSAL3)]:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    batch, T = int(g["batch"]), int(g["T"])
    for width in (128, 64, 128, 64):
        ops.set_gemm_tile(width)
        model = load_state(build(cfg), cases.e2e_state(cfg))
        mixture, sources = cases.e2e_batch(cfg, batch, T, 0)
        out = model(mixture.to(dev))
        loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev))
        loss.backward(); torch.cuda.synchronize()
        devs = []
        for k, p in model.named_parameters():
            if p.numel() == 1: continue
            cs, ref = cases.checksum(p.grad), g["gradsum." + k]
            devs.append((abs(cs[1] - ref[1]) / ref[1], abs(cs[0] - ref[0]) / ref[1], k))
        devs.sort(reverse=True)
        print(name, "width", width, "loss dev %.2e" % abs(float(loss) - float(g["loss"])),
              " worst abs-sum:", ["%s %.1e" % (k.replace("separator.", ""), a) for a, b, k in devs[:4]],
              " worst sum: %.1e" % max(b for a, b, k in devs), flush=True)
