"""Times the pointwise GEMM with prepared weights against the in-kernel split on the training step's shapes (plain
epilogue; CTN_LIB selects a diagnostic build).  usage: [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
T = 3999
out = []
for (M, K) in [(512, 128), (512, 256), (256, 512), (128, 512)]:
    W = torch.randn(M, K, device=dev)
    A = W.t().contiguous()
    Ap = ops.prepared(K, M, [(W, True, 0, 0)], dev)
    X = ops.new_act(B, K, T, dev).normal_()
    Y = ops.new_act(B, M, T, dev)
    bias = torch.randn(M, device=dev)
    res = []
    for prep, width in ((0, 0), (1, 128), (1, 64), (2, 128), (3, 0)):
        kw = dict(bias=bias, tile_width=width, A_prep=Ap if prep else None, prep_kernel=prep)
        for _ in range(3):
            ops.gemm_nn(M, T, K, B, A, M, X, Y, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_nn(M, T, K, B, A, M, X, Y, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 10
        if prep == 0:
            Yref = Y.clone()
        same = "" if torch.equal(Y, Yref) else " DIFFERS"
        res.append("%s%s %.0fus %.0fTF%s" % (("split", "wave", "block", "wave8x256")[prep], width or "", us,
                                            2.0 * M * K * T * B / us * 1e-6, same))
    out.append("M%dK%d: " % (M, K) + ", ".join(res))
print(os.environ.get("CTN_LIB", "base").split("/")[-1])
print("\n".join(out))
