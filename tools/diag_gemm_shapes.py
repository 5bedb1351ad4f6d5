"""Per-shape time of the GEMM launches in one eager training step (paper config): which launches are the slow ones.
usage: [batch]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import convtasnet_oracle as O
from bench import build_model
from speech_amd import ops
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
from speech_amd.trainer import FlatTrainer
from speech_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model = build_model(O.CONFIG_PAPER, dev)
tr = FlatTrainer(model, PIT1d(NegSISDR(), 2))
mix, src = synth.make_batch(B, 2, 32000, 0)
mix, src = mix.to(dev), src.to(dev)
for _ in range(2):
    tr.step(mix, src)
rec = []
orig_nn, orig_nt = ops.gemm_nn, ops.gemm_nt


def nn(M, T, K, batch, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_nn(M, T, K, batch, *a, **k); e1.record()
    rec.append((("nn", M, K, k.get("xflags", 0), k.get("emode", 0)), 2.0 * M * T * K * batch, e0, e1))


def nt(M, N, T, batch, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_nt(M, N, T, batch, *a, **k); e1.record()
    rec.append((("nt", M, N, k.get("aflags", 0), k.get("bflags", 0)), 2.0 * M * N * T * batch, e0, e1))


ops.gemm_nn, ops.gemm_nt = nn, nt
import speech_amd.functional as Fn
tr.step(mix, src)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, work, e0, e1 in rec:
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += work
tot = sum(a[1] for a in agg.values())
print("batch", B, "GEMM total %.2f ms" % (tot / 1e3))
for key, (n, us, work) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-28s x%-3d avg %7.1f us  %6.1f TF  %5.1f%% of GEMM time" % (key, n, us / n, work / us * 1e-6, 100 * us / tot))
