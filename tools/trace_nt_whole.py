"""Cycle stamps of block (0, 0, 0) of a trace build of the whole-output weight-gradient kernel (-DWH_TRACE, loaded
through CTN_LIB): per wave and stage the cycles spent in its parts.  usage: heads|bneck [batch]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from speech_amd import ops, _lib
form = sys.argv[1] if len(sys.argv) > 1 else "heads"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
T, H, Bc, Sc = 3999, 512, 128, 128
act = lambda C: ops.new_act(B, C, T, dev).normal_()
y2, dout, dskip, dn1, y1, x = act(H), act(Bc), act(Sc), act(H), act(H), act(Bc)
g, be = torch.randn(H, device=dev), torch.randn(H, device=dev)
alpha = torch.tensor([0.25], device=dev)
st = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
st.c1, st.c2 = torch.randn(B, device=dev) * 1e-3, torch.randn(B, device=dev) * 1e-3
dWo, dWs, dW1 = torch.zeros(Bc, H, device=dev), torch.zeros(Sc, H, device=dev), torch.zeros(H, Bc, device=dev)
dbo, dbs, db1 = torch.zeros(Bc, device=dev), torch.zeros(Sc, device=dev), torch.zeros(H, device=dev)
da = torch.zeros(ops._lib.STAT_SLOTS, device=dev, dtype=torch.float64)
for _ in range(3):
    if form == "heads":
        ops.gemm_nt(Bc + Sc, H, T, B, dout, y2, dWo, H, Ab=dskip, M1=Bc, C2=dWs, ldc2=H, a_rowsum=dbo, a_rowsum2=dbs,
                    bflags=ops.XF_PRELU | ops.XF_NORM, b_alpha=alpha, b_stat=st, b_gamma=g, b_beta=be)
    else:
        ops.gemm_nt(H, Bc, T, B, dn1, x, dW1, Bc, aflags=ops.XF_NORMBWD, A2=y1, a_alpha=alpha, a_stat=st, a_gamma=g,
                    a_rowsum=db1, a_dalpha=da)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((8, 192), dtype=np.uint64)
assert lib.ctn_debug_whole_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
tr = buf.astype(np.int64)
t0 = tr[:, 0].min()
for w in range(8):
    n = int((tr[w] > 0).sum())
    s = tr[w, :n]
    d = np.diff(s)
    k = (n - 2) // 4
    print("wave %d: %d stamps, first %d, loop %d cycles (%d per stage), tail %d" % (w, n, s[0] - t0, s[4 * k] - s[0], (s[4 * k] - s[0]) // max(k, 1), s[n - 1] - s[4 * k]))
    for i in range(8, 14):
        print("   stage %d:" % i, " ".join("%5d" % v for v in d[4 * i:4 * i + 4]))
    if w in (0, 4):
        print("   stage totals:", " ".join("%d" % v for v in d[:4 * k].reshape(k, 4).sum(axis=1)))
    med = np.median(d[:4 * k].reshape(k, 4), axis=0)
    print("   median parts:", " ".join("%5d" % v for v in med))
