"""Durations of the four codec launches (encoder forward / backward, mask * decoder forward / backward) at the paper
configuration's shapes (N = 512, L = 16, S = 8, 2 sources, 4 s at 8 kHz), with the bytes each moves.  usage: [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
N, L, S, NS, Tin = 512, 16, 8, 2, 32000
Tf = (Tin - L) // S + 1
x = torch.randn(B, Tin, device=dev)
We, Wd = torch.randn(N, 1, L, device=dev) * 0.1, torch.randn(N, 1, L, device=dev) * 0.1
act = lambda C: ops.new_act(B, C, Tf, dev).normal_()
w, mask, dz, dwd, dwa, dn = act(N), act(NS * N).sigmoid_(), act(NS * N), act(N), act(N), act(N)
out, dout = torch.empty(B, NS, Tin, device=dev), torch.randn(B, NS, Tin, device=dev)
dWe, dWd = torch.zeros_like(We), torch.zeros_like(Wd)
g, dg, db = torch.randn(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
st = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
st.c1, st.c2 = torch.randn(B, device=dev) * 1e-3, torch.randn(B, device=dev) * 1e-3
acc = ops.Accum(B, Tf, False, dev)
MB = 4 * B * N * Tf / 1e6
cases = [
    ("encoder_fwd", lambda: ops.encoder_fwd(x, 0, We, S, True, w, acc), MB),
    ("maskdecode_fwd", lambda: ops.maskdecode_fwd(w, mask, NS, Wd, S, out, 0), 3 * MB),
    ("maskdecode_bwd", lambda: ops.maskdecode_bwd(dout, 0, w, mask, NS, Wd, S, True, dz, dwd, dWd), 6 * MB),
    ("encoder_bwd", lambda: ops.encoder_bwd(x, 0, N, L, S, True, w, dWe, dw_a=dwa, dn=dn, stat=st, gamma=g, dgamma=dg,
                                            dbeta=db), 3 * MB),
]
tot = 0.0
for name, fn, mb in cases:
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print("%-16s %7.1f us  %7.1f MB  %5.2f TB/s" % (name, us, mb, mb / us))
print("total %.1f us" % tot)
