"""Training-step time of DPRNN-TasNet at the reference recipe's defaults (N=64, L=2, F=64, H=128, K=250, P=125, B=6,
4 s at 16 kHz): forward + PIT SI-SDR + backward + clip 5 + Adam, with a per-kernel-family split from torch's profiler.
usage: [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dprnn_cases as cases
from speech_amd.models.dprnn_tasnet import DPRNNTasNet
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
from speech_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
torch.manual_seed(111)
model = DPRNNTasNet(**cases.PAPER).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
crit = PIT1d(NegSISDR(), 2)
mix, src = synth.make_batch(B, 2, 64000, 0)
mix, src = mix.to(dev), src.to(dev)


def step():
    loss, _ = crit(model(mix), src)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("DPRNN-TasNet batch %d: %.1f ms/step, %.1f utt/s, loss %.4f" % (B, dt * 1e3, B / dt, loss.item()))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]
tot = sum(e.device_time_total for e in prof.key_averages())
for e in rows:
    print("%5.1f%% %8.2f ms  x%-4d %s" % (100 * e.device_time_total / tot, e.device_time_total / 1e3, e.count, e.key[:90]))
