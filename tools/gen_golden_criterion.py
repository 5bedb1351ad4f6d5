"""Generate tests/golden/criterion.npz by running the REFERENCE's own criterion modules on CPU: plain SDR, PIT over
it, the 'sum' reduction, one-and-rest PIT and Sinkhorn PIT.  Runs only in the build container (needs /root/reference);
the fixture holds plain arrays: inputs, the reference's losses / indices / matrices and its gradients.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_criterion.py
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from criterion.sdr import SDR, NegSDR, SISDR, NegSISDR  # noqa: E402  (reference)
from criterion.pit import PIT1d, ORPIT, SinkPIT  # noqa: E402  (reference)

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "speech_amd", "utils", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.path.join(ROOT, "tests", "golden", "criterion.npz")
T = 1000
CRITERIA = {"SDR": SDR, "NegSDR": NegSDR, "SISDR": SISDR, "NegSISDR": NegSISDR}


def np_(t):
    return t.detach().cpu().numpy().copy()


def signals(B, S, seed, mix=0.8, noise=0.03):
    """Targets and estimates that are rolled, scaled, noisy copies of them (so the values are realistic SDRs)."""
    tgt = synth.normal((B, S, T), seed, std=0.1)
    perm = [list(np.roll(np.arange(S), k % S)) for k in range(B)]
    est = torch.stack([tgt[i, perm[i]] for i in range(B)]) * mix + synth.normal((B, S, T), seed + 1, std=noise)
    return est, tgt


def main():
    d = {"meta": np.array("reference run on torch {} CPU fp32 (reference pins torch 1.10)".format(torch.__version__))}

    # --- plain SDR criteria: input ranks 2, 3, 4; reductions; batch_mean
    est3, tgt3 = signals(3, 3, 201)
    d["sdr_est"], d["sdr_tgt"] = np_(est3), np_(tgt3)
    for cname in ("SDR", "NegSDR"):
        for red in ("mean", "sum", None):
            for bm in (True, False):
                e = est3.clone().requires_grad_(True)
                loss = CRITERIA[cname](reduction=red)(e, tgt3, batch_mean=bm)
                loss.sum().backward()
                key = "sdr_{}_{}_{}".format(cname, red, int(bm))
                d[key + "_loss"], d[key + "_dest"] = np_(loss), np_(e.grad)
    e2 = est3[:, 0].clone().requires_grad_(True)                       # (B, T)
    loss = NegSDR()(e2, tgt3[:, 0], batch_mean=False)
    loss.sum().backward()
    d["sdr_rank2_loss"], d["sdr_rank2_dest"] = np_(loss), np_(e2.grad)
    est4, tgt4 = signals(2, 6, 211)                                     # viewed as (B, 3 sources, 2 microphones, T)
    est4, tgt4 = est4.view(2, 3, 2, T), tgt4.view(2, 3, 2, T)
    d["sdr4_est"], d["sdr4_tgt"] = np_(est4), np_(tgt4)
    e4 = est4.clone().requires_grad_(True)
    loss = NegSDR()(e4, tgt4, batch_mean=False)
    loss.sum().backward()
    d["sdr4_loss"], d["sdr4_dest"] = np_(loss), np_(e4.grad)

    # --- PIT over criteria other than NegSISDR/mean
    for S in (2, 3):
        est, tgt = signals(4, S, 220 + S)
        d["pit{}_est".format(S)], d["pit{}_tgt".format(S)] = np_(est), np_(tgt)
        for cname, red in (("NegSDR", "mean"), ("SDR", "mean"), ("NegSISDR", "sum"), ("SISDR", "mean"), ("NegSDR", "sum")):
            e = est.clone().requires_grad_(True)
            crit = PIT1d(CRITERIA[cname](reduction=red), S)
            loss_b, perm = crit(e, tgt, batch_mean=False)
            loss, _ = crit(e, tgt)
            loss.backward()
            key = "pit{}_{}_{}".format(S, cname, red)
            d[key + "_loss_b"], d[key + "_loss"], d[key + "_perm"], d[key + "_dest"] = np_(loss_b), np_(loss), np_(perm), np_(e.grad)

    # --- one-and-rest PIT: tensor target (3 sources) and packed target (3 and 2 sources)
    _, tgt = signals(3, 3, 231)
    one = [1, 0, 2]
    est = torch.stack([torch.stack([tgt[b, one[b]], tgt[b].sum(0) - tgt[b, one[b]]]) for b in range(3)])
    est = est * 0.9 + synth.normal((3, 2, T), 233, std=0.02)
    d["orpit_est"], d["orpit_tgt"] = np_(est), np_(tgt)
    for cname in ("NegSISDR", "SISDR", "NegSDR"):
        e = est.clone().requires_grad_(True)
        crit = ORPIT(CRITERIA[cname]())
        loss_b, idx = crit(e, tgt, batch_mean=False)
        loss, _ = crit(e, tgt)
        loss.backward()
        key = "orpit_" + cname
        d[key + "_loss_b"], d[key + "_loss"], d[key + "_idx"], d[key + "_dest"] = np_(loss_b), np_(loss), np_(idx), np_(e.grad)
    lens = [3, 2]
    seqs = [tgt[0, :3], tgt[1, :2]]
    packed = nn.utils.rnn.pack_sequence(seqs, enforce_sorted=True)
    est_p = torch.stack([torch.stack([tgt[0, 2], tgt[0, 0] + tgt[0, 1]]), torch.stack([tgt[1, 1], tgt[1, 0]])])
    est_p = est_p * 0.9 + synth.normal((2, 2, T), 235, std=0.02)
    d["orpit_packed_est"], d["orpit_packed_lens"] = np_(est_p), np.array(lens)
    e = est_p.clone().requires_grad_(True)
    crit = ORPIT(NegSISDR())
    loss_b, idx = crit(e, packed, batch_mean=False)
    loss, _ = crit(e, packed)
    loss.backward()
    d["orpit_packed_loss_b"], d["orpit_packed_loss"], d["orpit_packed_idx"], d["orpit_packed_dest"] = np_(loss_b), np_(loss), np_(idx), np_(e.grad)

    # --- Sinkhorn PIT
    for S in (2, 3, 4):
        est, tgt = signals(3, S, 240 + S)
        d["sink{}_est".format(S)], d["sink{}_tgt".format(S)] = np_(est), np_(tgt)
        for cname, cold, it in (("NegSISDR", 1.0, 10), ("NegSISDR", 0.2, 10), ("SISDR", 1.0, 5), ("NegSDR", 0.5, 10)):
            e = est.clone().requires_grad_(True)
            crit = SinkPIT(CRITERIA[cname](), n_sources=S, coldness=cold, iteration=it)
            loss_b, pattern = crit(e, tgt, batch_mean=False)
            loss, _ = crit(e, tgt)
            loss.backward()
            key = "sink{}_{}_{}_{}".format(S, cname, cold, it)
            d[key + "_loss_b"], d[key + "_loss"], d[key + "_pattern"], d[key + "_dest"] = np_(loss_b), np_(loss), np_(pattern), np_(e.grad)

    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(d), "arrays")


if __name__ == "__main__":
    main()
