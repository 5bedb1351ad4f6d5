"""Training-step time of a named oracle configuration (paper | causal3). usage: config [batch] [streams] [graph]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import convtasnet_oracle as O
from speech_amd.models.conv_tasnet import ConvTasNet
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
from speech_amd.trainer import FlatTrainer
from speech_amd.utils import synth
name = sys.argv[1] if len(sys.argv) > 1 else "paper"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 1
graph = len(sys.argv) > 4 and sys.argv[4] == "graph"
cfg = {"paper": O.CONFIG_PAPER, "causal3": O.CONFIG_CAUSAL3}[name]
dev = torch.device("cuda:0")
model = ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis="trainable", dec_basis="trainable",
                   enc_nonlinear=cfg["enc_nonlinear"], sep_hidden_channels=cfg["sep_hidden_channels"],
                   sep_bottleneck_channels=cfg["sep_bottleneck_channels"], sep_skip_channels=cfg["sep_skip_channels"],
                   sep_kernel_size=cfg["sep_kernel_size"], sep_num_blocks=cfg["sep_num_blocks"],
                   sep_num_layers=cfg["sep_num_layers"], dilated=True, separable=True, sep_nonlinear="prelu",
                   sep_norm=True, mask_nonlinear=cfg["mask_nonlinear"], causal=cfg["causal"], n_sources=cfg["n_sources"]).to(dev)
synth.fill_parameters(model, 1)
tr = FlatTrainer(model, PIT1d(NegSISDR(), cfg["n_sources"]), lr=1e-3, max_norm=5.0, n_streams=streams, use_graph=graph)
mix, src = synth.make_batch(B, cfg["n_sources"], 32000, 0)
mix, src = mix.to(dev), src.to(dev)
for _ in range(5):
    tr.step(mix, src)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 8
for _ in range(n):
    tr.step(mix, src)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("%s batch %d streams %d graph %s: %.2f ms/step, %.1f utt/s" % (name, B, streams, graph, dt * 1e3, B / dt))
