"""Depthwise forward/backward on very short rows against torch (diagnostic)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from speech_amd.functional import DepthwiseFn
dev = torch.device("cuda:0")
torch.manual_seed(0)
for T in (1, 2, 3, 4, 5, 7, 9):
    for d in (1, 2, 4, 8):
        C, B, P = 6, 3, 3
        x = torch.randn(B, C, T, dtype=torch.float64, requires_grad=True)
        w = torch.randn(C, 1, P, dtype=torch.float64, requires_grad=True)
        b = torch.randn(C, dtype=torch.float64, requires_grad=True)
        ref = F.conv1d(F.pad(x, (d, d)), w, b, dilation=d, groups=C)
        gy = torch.randn_like(ref)
        ref.backward(gy)
        xg = x.detach().float().to(dev).requires_grad_(True)
        wg = w.detach().float().to(dev).requires_grad_(True)
        bg = b.detach().float().to(dev).requires_grad_(True)
        y = DepthwiseFn.apply(xg, wg, bg, d, d, d)
        y.backward(gy.float().to(dev))
        torch.cuda.synchronize()
        e = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30))
        print("T=%d d=%d  y %.2e dx %.2e dw %.2e db %.2e" % (T, d, e(y, ref), e(xg.grad, x.grad), e(wg.grad, w.grad), e(bg.grad, b.grad)))
