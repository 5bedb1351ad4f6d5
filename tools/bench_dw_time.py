"""Times the fused depthwise forward and backward kernels at the paper shape for every dilation of a TCN block.
    python tools/bench_dw_time.py [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
E = B * H * T * 4
RC = os.environ.get("DW_RC", "1") == "1"


def fwd(d):
    ops.dw_fwd(x, w, bias, y, d, d, xflags=3, alpha_in=alpha, stat=st, gamma=g, beta=be, alpha_out=alpha, acc=acc)


def bwd(d):
    ops.dw_bwd(dn2, x, w, dw, db, dx, d, d, xflags=3, alpha1=alpha, stat1=st, gamma1=g, beta1=be, dgamma1=dg1,
               dbeta1=db1, iacc=iacc, y2=y, alpha2=alpha, stat2=st, gamma2=g, dgamma2=dg2, dbeta2=db2, dalpha2=da2, bias=bias,
               recompute_y2=RC)


def timeit(fn, d, reps=20):
    for _ in range(3):
        fn(d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(d)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for d in [int(v) for v in os.environ.get("DW_DILS", "1,2,4,8,16,32,64,128").split(",")]:
    tf, tb = timeit(fwd, d), timeit(bwd, d)
    print("d=%3d  fwd %6.1f us %6.0f GB/s (2E)   bwd %6.1f us  %6.0f GB/s (3E) %6.0f GB/s (4E)" % (
        d, tf, 2 * E / tf / 1e3, tb, 3 * E / tb / 1e3, 4 * E / tb / 1e3), flush=True)
