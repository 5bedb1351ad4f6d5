"""Numerical study (CPU): what the 1x1 convolutions of the paper configuration lose if their operands are split into bf16
pieces and only the leading piece products are kept (the way a bf16 MFMA would compute them, with fp32 accumulation
emulated in float64).  Prints the waveform / loss error of the whole forward pass against float64 for
fp32 arithmetic, 2 pieces x 3 products and 3 pieces x 6 products.  usage: [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import cases
import convtasnet_oracle as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
torch.set_num_threads(8)
cfg = O.CONFIG_PAPER
sd = cases.e2e_state(cfg)
mixture, sources = cases.e2e_batch(cfg, 1, T, 3)
real_conv1d = F.conv1d


def pieces(x, n):
    out, r = [], x.float()
    for _ in range(n):
        p = r.to(torch.bfloat16).float()
        out.append(p)
        r = r - p
    return out


def make_conv(n_pieces, keep):
    def conv1d(x, w, b=None, *a, **k):
        if w.shape[-1] != 1 or k.get("groups", 1) != 1:
            return real_conv1d(x, w, b, *a, **k)
        xp, wp = pieces(x, n_pieces), pieces(w, n_pieces)
        y = 0
        for i in range(n_pieces):
            for j in range(n_pieces):
                if i + j < keep:       # keep = 2: hh, hl, lh ; keep = 3 with 3 pieces: the six leading products
                    y = y + real_conv1d(xp[j].double(), wp[i].double())
        y = y.float()
        return y if b is None else y + b.view(1, -1, 1)
    return conv1d


def run(dtype, conv=None):
    F.conv1d = conv or real_conv1d
    try:
        s = {k: v.to(dtype) for k, v in sd.items()}
        out = O.conv_tasnet_forward(mixture.to(dtype), s, cfg)
        loss, perm = O.pit_neg_sisdr(out, sources.to(dtype))
    finally:
        F.conv1d = real_conv1d
    return out.double(), loss.double(), perm


ref_out, ref_loss, ref_perm = run(torch.float64)
for name, args in (("fp32", (torch.float32, None)), ("bf16 2 pieces, 3 products", (torch.float32, make_conv(2, 2))),
                   ("bf16 3 pieces, 6 products", (torch.float32, make_conv(3, 3)))):
    out, loss, perm = run(*args)
    e = ((out - ref_out).abs().max() / ref_out.abs().max()).item()
    print("%-28s waveform rel err %.2e   loss rel err %.2e   perm equal %s" % (
        name, e, abs((loss - ref_loss) / ref_loss).item(), torch.equal(perm, ref_perm)), flush=True)
