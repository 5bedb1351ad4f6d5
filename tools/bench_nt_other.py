"""Durations of the step's three other weight-gradient launches through ctn_gemm_nt: the mask convolution (1024 x 128,
PReLU on load), the entry bottleneck (128 x 512, norm on load) and the last layer's single head (128 x 512, PReLU + norm
on load); CTN_NT_WHOLE=0 in the environment selects the tile-per-block kernels.  usage: [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
T = 3999
act = lambda C: ops.new_act(B, C, T, dev).normal_()
dmask, skip, dbn, enc, dskip, y2 = act(1024), act(128), act(128), act(512), act(128), act(512)
g, be = torch.randn(512, device=dev), torch.randn(512, device=dev)
alpha = torch.tensor([0.25], device=dev)
st = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
dWm, dWe, dWs = torch.zeros(1024, 128, device=dev), torch.zeros(128, 512, device=dev), torch.zeros(128, 512, device=dev)
dbm, dbe, dbs = torch.zeros(1024, device=dev), torch.zeros(128, device=dev), torch.zeros(128, device=dev)


def mask():
    ops.gemm_nt(1024, 128, T, B, dmask, skip, dWm, 128, a_rowsum=dbm, bflags=ops.XF_PRELU, b_alpha=alpha)


def entry():
    ops.gemm_nt(128, 512, T, B, dbn, enc, dWe, 512, a_rowsum=dbe, bflags=ops.XF_NORM, b_stat=st, b_gamma=g, b_beta=be)


def single():
    ops.gemm_nt(128, 512, T, B, dskip, y2, dWs, 512, a_rowsum=dbs, bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=alpha,
                b_stat=st, b_gamma=g, b_beta=be)


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


print("CTN_NT_WHOLE=%s batch %d: mask %.1f us, entry %.1f us, single head %.1f us" % (
    os.environ.get("CTN_NT_WHOLE", "1"), B, timeit(mask), timeit(entry), timeit(single)))
