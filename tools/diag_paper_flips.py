"""Paper-width network: per layer, how many PReLU inputs have a different sign than in a float64 run -- for this build (GPU)
and for the fp32 CPU oracle -- and how close to zero the flipped inputs are.  usage: [T]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import cases, convtasnet_oracle as O
import test_parity_gpu as tp
from speech_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
cfg = O.CONFIG_PAPER
sd = cases.e2e_state(cfg)
mix, src = cases.e2e_batch(cfg, 1, T, 11)


def oracle_pre(dtype):
    rec = []
    orig = O.layer_norm
    def ln(x, sdd, prefix, causal, eps=1e-12):
        rec.append((prefix, x.detach().clone()))       # x = prelu(pre-activation): same sign as the pre-activation
        return orig(x, sdd, prefix, causal, eps)
    O.layer_norm = ln
    with torch.no_grad():
        O.conv_tasnet_forward(mix.to(dtype), {k: v.to(dtype) for k, v in sd.items()}, cfg)
    O.layer_norm = orig
    return dict(rec)


u64, u32 = oracle_pre(torch.float64), oracle_pre(torch.float32)
cap = []
orig_dw_fwd = ops.dw_fwd
def dw_fwd(x, w, bias, y, *a, **k):
    orig_dw_fwd(x, w, bias, y, *a, **k)
    cap.append((x.clone(), y.clone()))
ops.dw_fwd = dw_fwd
model = tp.load_state(tp.build(cfg), sd)
with torch.no_grad():
    model(mix.cuda())
torch.cuda.synchronize()
names = ["separator.tdcn.net.%d.net.%d." % (r, l) for r in range(3) for l in range(8)]
tot = [0, 0, 0, 0]
print("%-30s %28s %28s" % ("layer", "y1 flips gpu / cpu32 (max |y| flipped)", "y2 flips gpu / cpu32"))
for name, (y1, y2) in zip(names, cap):
    r1, r2 = u64[name + "norm1d."], u64[name + "separable_conv1d.norm1d."]
    c1, c2 = u32[name + "norm1d."], u32[name + "separable_conv1d.norm1d."]
    y1, y2 = y1.cpu()[..., :r1.shape[-1]], y2.cpu()[..., :r2.shape[-1]]
    f1g, f1c = (y1 > 0) != (r1 > 0), (c1 > 0) != (r1 > 0)
    f2g, f2c = (y2 > 0) != (r2 > 0), (c2 > 0) != (r2 > 0)
    m1 = float(y1[f1g].abs().max()) if f1g.any() else 0.0
    m2 = float(y2[f2g].abs().max()) if f2g.any() else 0.0
    e1 = float((y1.double() - torch.where(r1 > 0, r1, r1 / 0.25)).abs().max()) if False else 0.0
    print("%-30s %6d / %-6d (%.1e)      %6d / %-6d (%.1e)" % (name[-16:], f1g.sum(), f1c.sum(), m1, f2g.sum(), f2c.sum(), m2))
    tot[0] += int(f1g.sum()); tot[1] += int(f1c.sum()); tot[2] += int(f2g.sum()); tot[3] += int(f2c.sum())
print("total flips: y1 gpu %d cpu32 %d | y2 gpu %d cpu32 %d" % tuple(tot))
