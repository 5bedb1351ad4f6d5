"""DPRNN-TasNet training step: device time per ATen operator and input shape (torch profiler, CPU+CUDA activities), to
find which tensor operations the element-wise copy / reduce kernels belong to.  usage: [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dprnn_cases as cases
from speech_amd.models.dprnn_tasnet import DPRNNTasNet
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
from speech_amd.utils import synth
from torch.profiler import profile, ProfilerActivity
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


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total)[:28]
for e in rows:
    print("%8.2f ms  x%-4d %-28s %s" % (e.self_device_time_total / 1e3, e.count, e.key[:28], str(e.input_shapes)[:110]), flush=True)
