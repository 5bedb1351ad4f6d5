"""The fused depthwise backward kernel on ONE set of buffers reused by every launch against a DIFFERENT set per launch
(24 sets, as the 24 layers of a training step use): how much of the difference between the isolated and the in-step
duration is the buffers being cold.  usage: [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
T, H, NSETS = 3999, 512, 24
act = lambda C: ops.new_act(B, C, T, dev).normal_()
vec = lambda n: torch.randn(n, device=dev)
sets = [(act(H), act(H), act(H)) for _ in range(NSETS)]            # x (= y1), dn2, dx per layer
w, bias = vec(H * 3).view(H, 3), vec(H)
st = ops.Stat(vec(B), vec(B).abs() + 0.5, 1, 0, False); st.c1, st.c2 = vec(B), vec(B)
alpha = torch.tensor([0.25], device=dev)
g, be = vec(H), vec(H)
iacc = ops.Accum(B, T, False, dev)
dw, db = torch.zeros(H, 3, device=dev), torch.zeros(H, device=dev)
dg1, db1, dg2, db2 = (torch.zeros(H, device=dev) for _ in range(4))
da2 = torch.zeros(32, device=dev, dtype=torch.float64)
E = B * H * T * 4


def bwd(i, d=4):
    x, dn2, dx = sets[i]
    ops.dw_bwd(dn2, x, w, dw, db, dx, d, d, xflags=3, alpha1=alpha, stat1=st, gamma1=g, beta1=be, dgamma1=dg1,
               dbeta1=db1, iacc=iacc, y2=x, alpha2=alpha, stat2=st, gamma2=g, dgamma2=dg2, dbeta2=db2, dalpha2=da2, bias=bias,
               recompute_y2=True)


def timeit(pick, reps=48):
    for i in range(NSETS):
        bwd(pick(i))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        bwd(pick(i))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_same, t_cycle = timeit(lambda i: 0), timeit(lambda i: i % NSETS)
print("dw_bwd batch %d: one buffer set %.1f us (%.0f GB/s), 24 sets in turn %.1f us (%.0f GB/s)" %
      (B, t_same, 3 * E / t_same / 1e3, t_cycle, 3 * E / t_cycle / 1e3))
