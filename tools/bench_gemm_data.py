"""Does the operand DATA change the GEMM rate?  Times the plain pointwise GEMM (M = K = 512) on all-zero, constant and
random operands: f32 MFMA power, and with it the clock the chip sustains, depends on how many bits toggle.  usage: [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
T, M, K = 3999, 512, 512
for name in ("zeros", "ones", "random"):
    A = torch.zeros(K, M, device=dev)
    X = ops.new_act(B, K, T, dev).zero_()
    if name == "ones":
        A.fill_(1.0); X.fill_(1.0)
    if name == "random":
        A.normal_(); X.normal_()
    Y = ops.new_act(B, M, T, dev)
    for _ in range(5):
        ops.gemm_nn(M, T, K, B, A, M, X, Y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.gemm_nn(M, T, K, B, A, M, X, Y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    print("%-7s %.1f us  %.1f TFLOP/s" % (name, us, 2.0 * M * K * T * B / us * 1e-6))
