"""Accuracy of the two routes to the norm-2 backward coefficients (statistics epilogue of the heads' input-gradient GEMM
against the by-products of the heads' weight-gradient launch) in the f32 and three-piece arithmetic: kink-free paper
network, batch 2, every gradient against a float64 run of the oracle.  usage: [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases
import convtasnet_oracle as O
from test_parity_gpu import build, load_state, dev
from speech_amd import ops, functional
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
cfg = O.CONFIG_PAPER
sd = cases.e2e_state(cfg)
for k in sd:
    if k.endswith("nonlinear1d.weight") or k.endswith("prelu.weight"):
        sd[k].fill_(1.0)
mixture, sources = cases.e2e_batch(cfg, 2, T, 7)
_, perm64, out64, g64 = O.loss_and_grads(mixture.double(), sources.double(), {k: v.double() for k, v in sd.items()}, cfg)
for route in (True, False):
    for split in (3, 0, 2):
        functional.HEADS_COEFS_FROM_WGRAD = route
        ops.set_gemm_split(split)
        model = load_state(build(cfg), sd)
        out = model(mixture.to(dev()))
        loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources.to(dev()))
        loss.backward()
        torch.cuda.synchronize()
        es, ev, worst = 0.0, 0.0, ""
        for k, p in model.named_parameters():
            ref = g64[k].to(p.grad.device)
            e = float((p.grad.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            if p.numel() == 1:
                if e > es:
                    es, worst = e, k
            else:
                ev = max(ev, e)
        print("coefficients from %s, arithmetic %s: scalars max rel err %.2e (%s), tensors %.2e" %
              ("weight gradient" if route else "epilogue", {3: "bf16x3", 0: "f32", 2: "bf16x2"}[split], es, worst, ev))
ops.set_gemm_split(3)
