"""Runs the fused depthwise forward and backward a few times at the paper shape (for rocprofv3 --pmc passes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
T, H = 3999, 512
act = lambda C: ops.new_act(B, C, T, dev).normal_()
vec = lambda n: torch.randn(n, device=dev)
x, y, dn2, dx = act(H), act(H), act(H), act(H)
w, bias = vec(H * 3).view(H, 3), vec(H)
st = ops.Stat(vec(B), vec(B).abs() + 0.5, 1, 0, False); st.c1, st.c2 = vec(B), vec(B)
alpha = torch.tensor([0.25], device=dev)
g, be = vec(H), vec(H)
acc = ops.Accum(B, T, False, dev); iacc = ops.Accum(B, T, False, dev)
dw, db = torch.zeros(H, 3, device=dev), torch.zeros(H, device=dev)
dg1, db1, dg2, db2 = (torch.zeros(H, device=dev) for _ in range(4))
da2 = torch.zeros(32, device=dev, dtype=torch.float64)
for _ in range(5):
    ops.dw_fwd(x, w, bias, y, d, d, xflags=3, alpha_in=alpha, stat=st, gamma=g, beta=be, alpha_out=alpha, acc=acc)
    ops.dw_bwd(dn2, x, w, dw, db, dx, d, d, xflags=3, alpha1=alpha, stat1=st, gamma1=g, beta1=be, dgamma1=dg1,
               dbeta1=db1, iacc=iacc, y2=y, alpha2=alpha, stat2=st, gamma2=g, dgamma2=dg2, dbeta2=db2, dalpha2=da2)
torch.cuda.synchronize()
