"""Host enqueue time vs GPU time of one training step (is the step launch-bound?). usage: [batch]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = O.CONFIG_PAPER
dev = torch.device("cuda:0")
model = ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis="trainable", dec_basis="trainable",
                   enc_nonlinear=cfg["enc_nonlinear"], sep_hidden_channels=cfg["sep_hidden_channels"],
                   sep_bottleneck_channels=cfg["sep_bottleneck_channels"], sep_skip_channels=cfg["sep_skip_channels"],
                   sep_kernel_size=cfg["sep_kernel_size"], sep_num_blocks=cfg["sep_num_blocks"],
                   sep_num_layers=cfg["sep_num_layers"], dilated=True, separable=True, sep_nonlinear="prelu",
                   sep_norm=True, mask_nonlinear=cfg["mask_nonlinear"], causal=cfg["causal"], n_sources=cfg["n_sources"]).to(dev)
synth.fill_parameters(model, 1)
tr = FlatTrainer(model, PIT1d(NegSISDR(), 2), lr=1e-3, max_norm=5.0)
mix, src = synth.make_batch(B, 2, 32000, 0)
mix, src = mix.to(dev), src.to(dev)
for _ in range(3):
    tr.step(mix, src)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    tr.step(mix, src)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("batch %d: host enqueue %.1f ms, until GPU done %.1f ms" % (B, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
