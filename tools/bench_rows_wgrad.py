"""Durations of ctn_rows_wgrad in both arithmetic forms at the dual-path blocks' shapes.  usage: [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops, _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 511104
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


for name, M, K in (("lstm W_ih gradient", 1024, 64), ("lstm W_hh gradient (one dir)", 512, 128), ("linear weight gradient", 64, 256)):
    dy = torch.randn(R, M, device=dev); x = torch.randn(R, K, device=dev)
    dW = torch.zeros(M, K, device=dev)
    f32 = timeit(lambda: ops.rows_wgrad(dy, x, dW=dW, arith=_lib.ARITH_F32))
    b3 = timeit(lambda: ops.rows_wgrad(dy, x, dW=dW, arith=_lib.ARITH_BF16X3))
    gf = 2.0 * R * K * M
    print("%-32s %dx%d  f32 %8.1f us (%5.1f TFLOP/s)  bf16x3 %8.1f us (%5.1f)  bytes %.0f MB" % (
        name, M, K, f32, gf / f32 / 1e6, b3, gf / b3 / 1e6, (R * K + R * M) * 4.0 / 1e6))
