"""Times the fused depthwise forward / backward with cumulative-norm (per-frame) statistics, causal padding.
usage: [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
T, H = 3999, 512
ld = ops.round4(T)
act = lambda C: ops.new_act(B, C, T, dev).normal_()
vec = lambda n: torch.randn(n, device=dev)
x, y, dn2, dx = act(H), act(H), act(H), act(H)
w, bias = vec(H * 3).view(H, 3), vec(H)
row = lambda: torch.randn(B, ld, device=dev)
st = ops.Stat(row(), row().abs() + 0.5, ld, 1, True); st.c1, st.c2 = row(), row()
alpha = torch.tensor([0.25], device=dev)
g, be = vec(H), vec(H)
dw, db = torch.zeros(H, 3, device=dev), torch.zeros(H, device=dev)
dg1, db1, dg2, db2 = (torch.zeros(H, device=dev) for _ in range(4))
da2 = torch.zeros(32, device=dev, dtype=torch.float64)
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for d in (1, 8, 128):
    acc = ops.Accum(B, T, True, dev); iacc = ops.Accum(B, T, True, dev)
    f = lambda: ops.dw_fwd(x, w, bias, y, d, 2 * d, xflags=3, alpha_in=alpha, stat=st, gamma=g, beta=be, alpha_out=alpha, acc=acc)
    bw = lambda: ops.dw_bwd(dn2, x, w, dw, db, dx, d, 2 * d, xflags=3, alpha1=alpha, stat1=st, gamma1=g, beta1=be, dgamma1=dg1,
                            dbeta1=db1, iacc=iacc, y2=y, alpha2=alpha, stat2=st, gamma2=g, dgamma2=dg2, dbeta2=db2, dalpha2=da2)
    print("cLN d=%3d  dw_fwd %.1f us  dw_bwd %.1f us" % (d, timeit(f), timeit(bw)), flush=True)
