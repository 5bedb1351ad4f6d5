"""Cycle stamps of one block of a trace build of ctn_heads_bwd (built with -DHB_TRACE): per wave, the cycles between the
stamps of a tile (start, after the n2 transform, then per k-tile: after the barrier / before the next one, epilogue start, end)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from speech_amd import ops, _lib
B = 32
dev = torch.device("cuda:0")
T, H, Bc, Sc = 3999, 512, 128, 128
act = lambda C: ops.new_act(B, C, T, dev).normal_()
y2, dout, dskip, dn2 = act(H), act(Bc), act(Sc), act(H)
Wo, Ws = torch.randn(Bc, H, device=dev) * 0.1, torch.randn(Sc, H, device=dev) * 0.1
g, be = torch.randn(H, device=dev), torch.randn(H, device=dev)
alpha = torch.tensor([0.25], device=dev)
st = ops.Stat(torch.randn(B, device=dev) * 0.1, torch.rand(B, device=dev) + 0.5, 1, 0, False)
Wp = ops.prepared(Bc + Sc, H, [(Wo, False, 0, 0), (Ws, False, Bc, 0)], dev)
dWo, dWs = torch.zeros(Bc, H, device=dev), torch.zeros(Sc, H, device=dev)
dbo, dbs = torch.zeros(Bc, device=dev), torch.zeros(Sc, device=dev)
acc = ops.Accum(B, T, False, dev)
for _ in range(3):
    ops.heads_bwd(dout, dskip, y2, alpha, st, g, be, Wp, dn2, acc, dWo, dWs, dbo, dbs)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((8, 4096), dtype=np.uint64)
assert lib.ctn_debug_hb_trace(ctypes.c_void_p(buf.ctypes.data)) == 0
EPI = int(os.environ.get('HB_TRACE_EPI', '0'))
per = 2 + 16 + 2 + 6 * EPI
for w in (0, 4):
    tr = buf[w].astype(np.int64)
    n = int((tr > 0).sum()) // per
    print("wave", w, "tiles", n)
    for ti in range(3, min(n, 7)):
        s = tr[ti * per:(ti + 1) * per + 1]
        d = np.diff(s)
        print("  tile %d: n2 %d | steps (work, barrier wait): %s | epilogue %d | total %d" % (
            ti, d[0], " ".join("%d+%d" % (d[2 + 2 * k], d[3 + 2 * k]) for k in range(8)), s[per - 1] - s[18], s[per - 1] - s[0]))
        print("     first-step lead-in %d" % d[1])
        if EPI:
            print("     epilogue pieces (Sg write, barrier, read+store, barrier)x2:", " ".join(str(x) for x in d[18:26]))
