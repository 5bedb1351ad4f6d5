"""Is the convolution output recomputed by the depthwise backward kernel the one the forward kernel stored?
With gamma2 = 0, c1 = 0, c2 = 1, mean2 = 0, alpha2 = 1 the kernel's dy2 IS its y2, so dx = conv^T(y2')."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
for (B, H, T, d) in ((2, 64, 124, 1), (2, 64, 124, 2), (2, 64, 3999, 1), (2, 64, 124, 8)):
    P = 3
    x = ops.new_act(B, H, T, dev).normal_()
    w = torch.randn(H, P, device=dev); bias = torch.randn(H, device=dev)
    g1 = torch.randn(H, device=dev); be1 = torch.randn(H, device=dev)
    a1 = torch.tensor([0.25], device=dev); one = torch.tensor([1.0], device=dev)
    st1 = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
    y2 = ops.new_act(B, H, T, dev)
    ops.dw_fwd(x, w, bias, y2, d, d, xflags=3, alpha_in=a1, stat=st1, gamma=g1, beta=be1)
    st2 = ops.Stat(torch.zeros(B, device=dev), torch.ones(B, device=dev), 1, 0, False)
    st2.c1 = torch.zeros(B, device=dev); st2.c2 = torch.ones(B, device=dev)
    zero_dn = ops.new_act(B, H, T, dev, zero=True)
    res = []
    for rc in (False, True):
        dw, db = torch.zeros(H, P, device=dev), torch.zeros(H, device=dev)
        z = [torch.zeros(H, device=dev) for _ in range(4)]
        da2 = torch.zeros(32, device=dev, dtype=torch.float64)
        iacc = ops.Accum(B, T, False, dev)
        dx = ops.new_act(B, H, T, dev)
        ops.dw_bwd(zero_dn, x, w, dw, db, dx, d, d, xflags=3, alpha1=a1, stat1=st1, gamma1=g1, beta1=be1,
                   dgamma1=z[0], dbeta1=z[1], iacc=iacc, y2=y2, alpha2=one, stat2=st2, gamma2=torch.zeros(H, device=dev),
                   dgamma2=z[2], dbeta2=z[3], dalpha2=da2, bias=bias, recompute_y2=rc)
        torch.cuda.synchronize()
        res.append((dx.clone(), db.clone(), da2.sum().item()))
    diff = (res[0][0] - res[1][0]).abs()
    print("B%d H%d T%d d%d: dx max diff %.3e (max |dx| %.3e) at %s; nonzero diffs %d of %d; dbias diff %.3e; dalpha %.9e vs %.9e" % (
        B, H, T, d, diff.max().item(), res[0][0].abs().max().item(), str(divmod(int(diff.argmax()), 1)), int((diff > 0).sum()),
        diff.numel(), (res[0][1] - res[1][1]).abs().max().item(), res[0][2], res[1][2]))
    idx = (diff > 0).nonzero()[:8].tolist()
    print("   first differing (b, c, t):", idx)

# second experiment: realistic backward (random dn2, gLN statistics of the real y2), slope gradient against float64
for (B, H, T, d) in ((2, 64, 124, 1), (2, 64, 124, 2), (2, 64, 124, 8)):
    P = 3
    x = ops.new_act(B, H, T, dev).normal_()
    dn2 = ops.new_act(B, H, T, dev).normal_()
    w = torch.randn(H, P, device=dev); bias = torch.randn(H, device=dev) * 0.1
    g1 = torch.randn(H, device=dev); be1 = torch.randn(H, device=dev) * 0.1; g2 = torch.randn(H, device=dev)
    a1 = torch.tensor([1.0], device=dev); a2 = torch.tensor([1.0], device=dev)
    st1 = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
    y2 = ops.new_act(B, H, T, dev)
    acc2 = ops.Accum(B, T, False, dev)
    ops.dw_fwd(x, w, bias, y2, d, d, xflags=3, alpha_in=a1, stat=st1, gamma=g1, beta=be1, alpha_out=a2, acc=acc2)
    st2 = ops.finalize_stats(acc2, B, H, T, 1e-12, dev)
    st2.c1 = torch.randn(B, device=dev) * 0.01; st2.c2 = torch.randn(B, device=dev) * 0.01
    y64 = y2.double().cpu()
    du = (g2.double().cpu().view(1, -1, 1) * dn2.double().cpu() * st2.rstd.double().cpu().view(-1, 1, 1)
          + st2.c1.double().cpu().view(-1, 1, 1) + (y64 - st2.mean.double().cpu().view(-1, 1, 1)) * st2.c2.double().cpu().view(-1, 1, 1))
    ref = (du * y64.clamp(max=0)).sum().item()
    out = []
    for rc in (False, True):
        dw, db = torch.zeros(H, P, device=dev), torch.zeros(H, device=dev)
        z = [torch.zeros(H, device=dev) for _ in range(4)]
        da2 = torch.zeros(32, device=dev, dtype=torch.float64)
        iacc = ops.Accum(B, T, False, dev)
        dx = ops.new_act(B, H, T, dev)
        ops.dw_bwd(dn2, x, w, dw, db, dx, d, d, xflags=3, alpha1=a1, stat1=st1, gamma1=g1, beta1=be1,
                   dgamma1=z[0], dbeta1=z[1], iacc=iacc, y2=y2, alpha2=a2, stat2=st2, gamma2=g2,
                   dgamma2=z[2], dbeta2=z[3], dalpha2=da2, bias=bias, recompute_y2=rc)
        torch.cuda.synchronize()
        out.append(da2.sum().item())
    print("d%d dalpha2: float64 %.9e stored %.9e recomputed %.9e" % (d, ref, out[0], out[1]))
