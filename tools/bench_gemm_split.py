"""One plain pointwise GEMM as a single launch vs two concurrent launches of different tile widths on two streams
(128-wide tiles on the first columns, 64-wide on the rest): do co-resident blocks of different durations take the
launch out of lockstep?  usage: [batch]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
T = 3999
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()
for (M, K) in [(512, 128), (512, 256), (256, 512), (128, 512)]:
    A = torch.randn(K, M, device=dev)
    X = ops.new_act(B, K, T, dev).normal_()
    Y = ops.new_act(B, M, T, dev)
    bias = torch.randn(M, device=dev)
    def single():
        ops.gemm_nn(M, T, K, B, A, M, X, Y, bias=bias)
    def split(T1):
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1):
            ops.gemm_nn(M, T1, K, B, A, M, X[:, :, :T1], Y[:, :, :T1], bias=bias, tile_width=128)
        with torch.cuda.stream(s2):
            ops.gemm_nn(M, T - T1, K, B, A, M, X[:, :, T1:], Y[:, :, T1:], bias=bias, tile_width=64)
        main.wait_stream(s1); main.wait_stream(s2)
    def timeit(fn):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / 20
    res = ["single %.1f" % timeit(single)]
    for T1 in (1024, 2048, 2688, 3072):
        res.append("T1=%d %.1f" % (T1, timeit(lambda: split(T1))))
    print("batch %d M%d K%d us:" % (B, M, K), " | ".join(res), flush=True)
