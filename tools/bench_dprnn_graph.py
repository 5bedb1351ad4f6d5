"""DPRNN-TasNet training step replayed from a HIP graph (forward + loss + backward captured, clip + Adam outside):
how much of the 590 ms eager step is launch overhead of MIOpen's per-time-step kernels.  usage: [batch]"""
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


def fwd_bwd():
    loss, _ = crit(model(mix), src)
    loss.backward()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        opt.zero_grad(set_to_none=False)
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=False)
with torch.cuda.graph(g):
    loss = fwd_bwd()
torch.cuda.synchronize()
print("captured", flush=True)


def step():
    for p in model.parameters():
        p.grad.zero_()
    g.replay()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("DPRNN-TasNet batch %d, graph replay: %.1f ms/step, %.1f utt/s, loss %.4f" % (B, dt * 1e3, B / dt, loss.item()))
