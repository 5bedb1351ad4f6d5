"""Reads the per-block phase stamps of a diagnostic build (tools/build_variants.sh STAMPS) after ONE launch of a
plain pointwise GEMM and prints how the blocks' phases line up.  usage: M K batch width"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from speech_amd import ops, _lib
M, K, B, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib = _lib.load()
ops.set_gemm_tile(W)
dev = torch.device("cuda:0"); T = 3999
A = torch.randn(K, M, device=dev); X = ops.new_act(B, K, T, dev).normal_(); Y = ops.new_act(B, M, T, dev)
bias = torch.randn(M, device=dev)
for _ in range(3):
    ops.gemm_nn(M, T, K, B, A, M, X, Y, bias=bias)
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
raw = ctypes.CDLL(os.environ["CTN_LIB"])
raw.ctn_debug_read_stamps.argtypes = [ctypes.c_void_p]
assert raw.ctn_debug_read_stamps(buf.ctypes.data) == 0
nblk = min(8192, ((M + 127) // 128) * ((T + W - 1) // W) * B)
s = buf.reshape(8192, 8)[:nblk].astype(np.int64)
t0 = s[:, 0].min()
rel = (s[:, :7] - t0) / 100.0     # s_memtime ticks at 100 MHz -> us
names = ["start", "issued", "tile0", "loop_end", "bar", "epi_end", "end"]
print("blocks", nblk, "kernel span %.1f us" % rel[:, 6].max())
for i, n in enumerate(names):
    print("%-9s min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f" % (n, rel[:, i].min(), np.median(rel[:, i]), np.percentile(rel[:, i], 90), rel[:, i].max()))
d = np.diff(rel, axis=1)
for i in range(6):
    print("phase %-9s->%-9s mean %6.2f us  p10 %6.2f  p90 %6.2f" % (names[i], names[i + 1], d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
hw = s[:, 7] if False else buf.reshape(8192, 8)[:nblk, 7]
cu = ((hw >> 32) << 8) | (((hw & 0xffffffff) >> 8) & 0xff)      # xcc, (se, cu) bits of HW_ID
print("distinct (xcc, se/cu) ids:", len(np.unique(cu)))
