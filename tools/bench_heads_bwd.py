"""Duration of the fused heads backward launch (ctn_heads_bwd) at the paper's layer shape, beside the two launches it
replaces (ctn_gemm_nn with the norm-backward epilogue + ctn_gemm_nt).  usage: [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
from speech_amd.ops import XF_PRELU, XF_NORM, EPI_NORMBWD
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
T, H, Bc, Sc = 3999, 512, 128, 128
act = lambda C: ops.new_act(B, C, T, dev).normal_()
y2, dout, dskip, dn2 = act(H), act(Bc), act(Sc), act(H)
Wo, Ws = torch.randn(Bc, H, device=dev) * 0.1, torch.randn(Sc, H, device=dev) * 0.1
g, be = torch.randn(H, device=dev), torch.randn(H, device=dev)
alpha = torch.tensor([0.25], device=dev)
st = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
Wp = ops.prepared(Bc + Sc, H, [(Wo, False, 0, 0), (Ws, False, Bc, 0)], dev)
dWo, dWs = torch.zeros(Bc, H, device=dev), torch.zeros(Sc, H, device=dev)
dbo, dbs = torch.zeros(Bc, device=dev), torch.zeros(Sc, device=dev)
acc = ops.Accum(B, T, False, dev)


def fused():
    ops.heads_bwd(dout, dskip, y2, alpha, st, g, be, Wp, dn2, acc, dWo, dWs, dbo, dbs)


def pair():
    ops.gemm_nn(H, T, Bc + Sc, B, Wo, H, dout, dn2, A2=Ws, lda2=H, K1=Bc, Xb=dskip, emode=EPI_NORMBWD, E=y2, ealpha=alpha,
                estat=st, egamma=g, eacc=acc, A_prep=Wp)
    ops.gemm_nt(Bc + Sc, H, T, B, dout, y2, dWo, H, Ab=dskip, M1=Bc, C2=dWs, ldc2=H, a_rowsum=dbo, a_rowsum2=dbs,
                bflags=XF_PRELU | XF_NORM, b_alpha=alpha, b_stat=st, b_gamma=g, b_beta=be)


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


tf, tp = timeit(fused), timeit(pair)
flops = 2 * 2.0 * (Bc + Sc) * H * T * B
print("batch %d: fused %.1f us (%.1f TFLOP/s algorithmic), two launches %.1f us (%.1f)" % (B, tf, flops / tf / 1e6, tp, flops / tp / 1e6))
