"""Durations of the residual layer's two weight-gradient launches (heads 256 x 512, bottleneck 512 x 128 with the
norm-backward operand) through ctn_gemm_nt; CTN_NT_WHOLE=1 in the environment selects the whole-output kernels.
usage: [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
T, H, Bc, Sc = 3999, 512, 128, 128
act = lambda C: ops.new_act(B, C, T, dev).normal_()
y2, dout, dskip, dn1, y1, x = act(H), act(Bc), act(Sc), act(H), act(H), act(Bc)
g, be = torch.randn(H, device=dev), torch.randn(H, device=dev)
alpha = torch.tensor([0.25], device=dev)
st = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
st.c1, st.c2 = torch.randn(B, device=dev) * 1e-3, torch.randn(B, device=dev) * 1e-3
dWo, dWs, dW1 = torch.zeros(Bc, H, device=dev), torch.zeros(Sc, H, device=dev), torch.zeros(H, Bc, device=dev)
dbo, dbs, db1 = torch.zeros(Bc, device=dev), torch.zeros(Sc, device=dev), torch.zeros(H, device=dev)
da = torch.zeros(ops._lib.STAT_SLOTS, device=dev, dtype=torch.float64)


def heads():
    ops.gemm_nt(Bc + Sc, H, T, B, dout, y2, dWo, H, Ab=dskip, M1=Bc, C2=dWs, ldc2=H, a_rowsum=dbo, a_rowsum2=dbs,
                bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=alpha, b_stat=st, b_gamma=g, b_beta=be)


def bneck():
    ops.gemm_nt(H, Bc, T, B, dn1, x, dW1, Bc, aflags=ops.XF_NORMBWD, A2=y1, a_alpha=alpha, a_stat=st, a_gamma=g,
                a_rowsum=db1, a_dalpha=da)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("CTN_NT_WHOLE=%s batch %d: heads %.1f us, bottleneck %.1f us" % (os.environ.get("CTN_NT_WHOLE", "0"), B, timeit(heads), timeit(bneck)))
