"""A few launches of ctn_rows_wgrad at one shape, for counter passes.  usage: M K [rows] [f32|b3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops, _lib
M, K = int(sys.argv[1]), int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 511104
arith = _lib.ARITH_F32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else _lib.ARITH_BF16X3
dev = torch.device("cuda:0")
dy = torch.randn(R, M, device=dev); x = torch.randn(R, K, device=dev); dW = torch.zeros(M, K, device=dev)
for _ in range(5):
    ops.rows_wgrad(dy, x, dW=dW, arith=arith)
torch.cuda.synchronize()
