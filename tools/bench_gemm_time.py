"""Times the plain pointwise GEMM on a few shapes (CTN_LIB selects a diagnostic build). usage: [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
T = 3999
out = []
for (M, K) in [(512, 128), (512, 512), (128, 512), (256, 512)]:
    A = torch.randn(K, M, device=dev)
    X = ops.new_act(B, K, T, dev).normal_()
    Y = ops.new_act(B, M, T, dev)
    bias = torch.randn(M, device=dev)
    for _ in range(3):
        ops.gemm_nn(M, T, K, B, A, M, X, Y, bias=bias)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_nn(M, T, K, B, A, M, X, Y, bias=bias)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    out.append("M%dK%d %.1fus %.1fTF" % (M, K, us, 2.0 * M * K * T * B / us * 1e-6))
print(os.environ.get("CTN_LIB", "base").split("/")[-1], " | ".join(out))
