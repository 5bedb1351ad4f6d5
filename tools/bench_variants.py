"""Training throughput of the reference README's other Conv-TasNet rows (egs/wsj0-mix/conv-tasnet/README.md: trainable +
ReLU / trainable, trainable / pseudo-inverse, Fourier / Fourier) through the module API: model(mixture) -> PIT SI-SDR ->
backward -> clip 5 -> torch.optim.Adam, eager launches, one GPU.  usage: [batch] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
from speech_amd.models.conv_tasnet import ConvTasNet
from speech_amd.utils import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
ROWS = {
    "trainable+ReLU / trainable": dict(enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu"),
    "trainable / trainable": dict(enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None),
    "trainable / pinv": dict(enc_basis="trainable", dec_basis="pinv", enc_nonlinear=None),
    "Fourier / Fourier (one-sided, real)": dict(enc_basis="Fourier", dec_basis="Fourier", enc_nonlinear=None,
                                                enc_onesided=True, enc_return_complex=False, window_fn="hann"),
}
mix, src = synth.make_batch(B, 2, 32000, 0)
mix, src = mix.to(dev), src.to(dev)
for name, kw in ROWS.items():
    torch.manual_seed(111)
    model = ConvTasNet(512, 16, stride=8, sep_hidden_channels=512, sep_bottleneck_channels=128, sep_skip_channels=128,
                       sep_kernel_size=3, sep_num_blocks=3, sep_num_layers=8, dilated=True, separable=True,
                       sep_nonlinear="prelu", sep_norm=True, mask_nonlinear="sigmoid", causal=False, n_sources=2,
                       **kw).to(dev)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    crit = PIT1d(NegSISDR(), 2)

    def step():
        loss, _ = crit(model(mix), src)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-40s %7.1f utt/s  %6.1f ms/step  fused=%s  loss %.3f" % (name, B / dt, dt * 1e3, model._fused_ok(), loss.item()), flush=True)
