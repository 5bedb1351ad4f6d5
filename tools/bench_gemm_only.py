"""Runs one plain pointwise GEMM shape a few times (for rocprofv3 --pmc passes). usage: M K [batch] [reps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
M, K = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
T = 3999
A = torch.randn(K, M, device=dev)
X = ops.new_act(B, K, T, dev).normal_()
Y = ops.new_act(B, M, T, dev)
bias = torch.randn(M, device=dev)
for _ in range(reps):
    ops.gemm_nn(M, T, K, B, A, M, X, Y, bias=bias)
torch.cuda.synchronize()
