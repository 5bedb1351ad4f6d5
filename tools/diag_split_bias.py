"""Systematic bias of the GEMM arithmetic modes: all-positive operands (no cancellation), relative error of every output
against float64, mean (bias) and rms.  usage: [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
dev = torch.device("cuda:0")
M, T, B = 256, 3999, 2
for K in [int(v) for v in (sys.argv[1:] or ["128", "512", "2048"])]:
    torch.manual_seed(0)
    W = torch.rand(M, K, device=dev) + 0.5
    X = ops.new_act(B, K, T, dev)
    X[:, :, :T] = torch.rand(B, K, T, device=dev) + 0.5
    ref = torch.einsum("mk,bkt->bmt", W.double(), X[:, :, :T].double())
    A = W.t().contiguous()
    for split in (0, 3, 2):
        ops.set_gemm_split(split)
        Y = ops.new_act(B, M, T, dev)
        ops.gemm_nn(M, T, K, B, A, M, X, Y)
        torch.cuda.synchronize()
        rel = (Y[:, :, :T].double() - ref) / ref
        print("K %5d  arithmetic %-7s mean rel err %+.3e   rms %.3e" % (K, {0: "f32", 3: "bf16x3", 2: "bf16x2"}[split],
                                                                         rel.mean().item(), rel.pow(2).mean().sqrt().item()))
ops.set_gemm_split(3)
