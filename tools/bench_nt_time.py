"""Times the weight-gradient GEMM on the paper configuration's launches (CTN_LIB selects a diagnostic build).
usage: [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
KC = [int(v) if int(v) > 0 else None for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [None, None, None]
dev = torch.device("cuda:0")
T = 3999


def stat():
    m = torch.zeros(B, device=dev); r = torch.ones(B, device=dev)
    s = ops.Stat(m, r, 1, 0, False)
    s.c1 = torch.zeros(B, device=dev); s.c2 = torch.zeros(B, device=dev)
    return s


def timed(fn, flops):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    return "%.1fus %.1fTF" % (us, flops / us * 1e-6)


alpha = torch.full((1,), 0.25, device=dev)
out = []
# output + skip heads: A = [dout; dskip] (128 + 128 rows), B = y2 (512 rows) with PReLU + norm on load
do, ds, y2 = (ops.new_act(B, n, T, dev).normal_() for n in (128, 128, 512))
C1, C2 = torch.zeros(128, 512, device=dev), torch.zeros(128, 512, device=dev)
r1, r2 = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
g, be = torch.ones(512, device=dev), torch.zeros(512, device=dev)
st = stat()
out.append("heads " + timed(lambda: ops.gemm_nt(256, 512, T, B, do, y2, C1, 512, Ab=ds, M1=128, C2=C2, ldc2=512, a_rowsum=r1,
                                                a_rowsum2=r2, bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=alpha, b_stat=st,
                                                b_gamma=g, b_beta=be, kchunk=KC[0]), 2.0 * 256 * 512 * T * B))
# bottleneck conv of a layer: A = dn1 (512 rows) with norm / PReLU backward on load, B = layer input (128 rows)
dn1, y1, x = (ops.new_act(B, n, T, dev).normal_() for n in (512, 512, 128))
W1 = torch.zeros(512, 128, device=dev)
b1 = torch.zeros(512, device=dev)
da = torch.zeros(ops._lib.STAT_SLOTS, device=dev, dtype=torch.float64)
out.append("w1 " + timed(lambda: ops.gemm_nt(512, 128, T, B, dn1, x, W1, 128, aflags=ops.XF_NORMBWD, A2=y1, a_alpha=alpha,
                                             a_stat=st, a_gamma=g, a_rowsum=b1, a_dalpha=da, kchunk=KC[1]), 2.0 * 512 * 128 * T * B))
# mask conv: A = dz (1024 rows), B = skip sum (128 rows) with PReLU on load
dz = ops.new_act(B, 1024, T, dev).normal_()
Wm = torch.zeros(1024, 128, device=dev)
bm = torch.zeros(1024, device=dev)
out.append("mask " + timed(lambda: ops.gemm_nt(1024, 128, T, B, dz, x, Wm, 128, a_rowsum=bm, bflags=ops.XF_PRELU, b_alpha=alpha,
                                               kchunk=KC[2]),
                           2.0 * 1024 * 128 * T * B))
out.append("plain " + timed(lambda: ops.gemm_nt(1024, 128, T, B, dz, x, Wm, 128, kchunk=KC[2]), 2.0 * 1024 * 128 * T * B))
out.append("plain+rowsum " + timed(lambda: ops.gemm_nt(1024, 128, T, B, dz, x, Wm, 128, a_rowsum=bm, kchunk=KC[2]), 2.0 * 1024 * 128 * T * B))
print(os.environ.get("CTN_LIB", "base").split("/")[-1], os.environ.get("CTN_NT_TILE", ""), KC, " | ".join(out))
