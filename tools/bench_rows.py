"""Durations of the row-product kernels at the dual-path blocks' shapes (recipe defaults, batch 4: 205 200 rows), beside
torch.matmul (the GEMM library) on the same tensors as a yardstick.  usage: [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 205200
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("%-34s %9s %9s %8s %8s" % ("product", "own us", "torch us", "GB/s", "TFLOP/s"))
for name, K, M in (("lstm input projection", 64, 1024), ("linear fwd (intra / inter)", 256, 64), ("linear fwd one direction", 128, 64)):
    x = torch.randn(R, K, device=dev); w = torch.randn(M, K, device=dev); b = torch.randn(M, device=dev)
    own = timeit(lambda: ops.rows_gemm(x, w, b))
    lib = timeit(lambda: torch.addmm(b, x, w.t()))
    byt = (R * K + R * M) * 4.0
    print("%-34s %9.1f %9.1f %8.0f %8.1f" % (name + " %dx%d" % (K, M), own, lib, byt / own / 1e3, 2.0 * R * K * M / own / 1e6))
for name, K, M in (("lstm input gradient", 1024, 64), ("linear input gradient", 64, 256)):
    dy = torch.randn(R, K, device=dev); w = torch.randn(K, M, device=dev)
    own = timeit(lambda: ops.rows_gemm(dy, w, w_is_km=True))
    lib = timeit(lambda: dy @ w)
    byt = (R * K + R * M) * 4.0
    print("%-34s %9.1f %9.1f %8.0f %8.1f" % (name + " %dx%d" % (K, M), own, lib, byt / own / 1e3, 2.0 * R * K * M / own / 1e6))
for name, M, K in (("lstm W_ih gradient", 1024, 64), ("lstm W_hh gradient (one dir)", 512, 128), ("linear weight gradient", 64, 256)):
    dy = torch.randn(R, M, device=dev); x = torch.randn(R, K, device=dev)
    dW = torch.zeros(M, K, device=dev)
    own = timeit(lambda: ops.rows_wgrad(dy, x, dW=dW))
    lib = timeit(lambda: dy.t() @ x)
    byt = (R * K + R * M) * 4.0
    print("%-34s %9.1f %9.1f %8.0f %8.1f" % (name + " %dx%d" % (M, K), own, lib, byt / own / 1e3, 2.0 * R * K * M / own / 1e6))
