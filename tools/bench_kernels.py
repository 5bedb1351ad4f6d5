"""Per-kernel micro-benchmark at the paper-config shapes (HIP-event timing, median of repeats)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = 3999
REP = 20


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REP):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e-3


def act(C):
    return ops.new_act(B, C, T, dev).normal_()


def vec(n, dt=torch.float32):
    return torch.randn(n, device=dev).to(dt)


print("batch", B)
# ---- gemm_nn shapes
for name, M, K, mode in [("bottleneck fwd  M512 K128 STATS", 512, 128, "stats"),
                         ("heads fwd       M256 K512 NORM+HEADS", 256, 512, "heads"),
                         ("sep bottleneck  M128 K512 plain", 128, 512, "plain"),
                         ("heads dgrad     M512 K256 NORMBWD-epi", 512, 256, "nbwd_epi"),
                         ("bneck dgrad     M128 K512 NORMBWD-load", 128, 512, "nbwd_load"),
                         ("mask fwd        M1024 K128 SIGMOID", 1024, 128, "sigmoid"),
                         ("mask dgrad      M128 K1024 PRELUBWD", 128, 1024, "plain"),
                         ("plain           M512 K128", 512, 128, "plain"),
                         ("plain           M512 K512", 512, 512, "plain")]:
    A = vec(K * M).view(K, M)
    X, Y = act(K), act(M)
    st = ops.Stat(vec(B), vec(B).abs() + 0.5, 1, 0, False)
    st.c1, st.c2 = vec(B), vec(B)
    alpha = torch.tensor([0.25], device=dev)
    g, be, bias = vec(max(M, K)), vec(max(M, K)), vec(M)
    if mode == "plain":
        fn = lambda: ops.gemm_nn(M, T, K, B, A, M, X, Y, bias=bias)
    elif mode == "stats":
        acc = ops.Accum(B, T, False, dev)
        fn = lambda: ops.gemm_nn(M, T, K, B, A, M, X, Y, bias=bias, emode=ops.EPI_STATS, ealpha=alpha, eacc=acc)
    elif mode == "heads":
        R, Y2 = act(128), act(128)
        fn = lambda: ops.gemm_nn(M, T, K, B, A, M, X, Y, xflags=3, alpha=alpha, stat=st, gamma=g, beta=be, bias=bias,
                                 bias2=bias, emode=ops.EPI_HEADS, Mo=128, R=R, Y2=Y2, accumulate=True)
    elif mode == "sigmoid":
        fn = lambda: ops.gemm_nn(M, T, K, B, A, M, X, Y, xflags=1, alpha=alpha, bias=bias, emode=ops.EPI_SIGMOID)
    elif mode == "nbwd_epi":
        E = act(M)
        acc = ops.Accum(B, T, False, dev)
        fn = lambda: ops.gemm_nn(M, T, K, B, A, M, X, Y, emode=ops.EPI_NORMBWD, E=E, ealpha=alpha, estat=st, egamma=g,
                                 eacc=acc)
    elif mode == "nbwd_load":
        X2, R = act(K), act(M)
        fn = lambda: ops.gemm_nn(M, T, K, B, A, M, X, Y, xflags=4, X2=X2, alpha=alpha, stat=st, gamma=g,
                                 emode=ops.EPI_ADD, R=R)
    dt = timeit(fn)
    fl = 2.0 * M * K * T * B
    print("gemm_nn %-40s %8.1f us  %6.1f TFLOP/s" % (name, dt * 1e6, fl / dt / 1e12))

# ---- gemm_nt shapes
for name, M, N, bwd in [("dW1   M512 N128 NORMBWD A", 512, 128, True), ("heads M256 N512 NORM B", 256, 512, False),
                        ("mask  M1024 N128", 1024, 128, False), ("sepb  M128 N512", 128, 512, False)]:
    A, Bm = act(M), act(N)
    Cw = torch.zeros(M, N, device=dev)
    st = ops.Stat(vec(B), vec(B).abs() + 0.5, 1, 0, False)
    st.c1, st.c2 = vec(B), vec(B)
    alpha = torch.tensor([0.25], device=dev)
    g = vec(max(M, N))
    rs = torch.zeros(M, device=dev)
    if bwd:
        A2 = act(M)
        da = torch.zeros(32, device=dev, dtype=torch.float64)
        fn = lambda: ops.gemm_nt(M, N, T, B, A, Bm, Cw, N, aflags=4, A2=A2, a_alpha=alpha, a_stat=st, a_gamma=g,
                                 a_rowsum=rs, a_dalpha=da)
    else:
        fn = lambda: ops.gemm_nt(M, N, T, B, A, Bm, Cw, N, a_rowsum=rs)
    dt = timeit(fn)
    fl = 2.0 * M * N * T * B
    print("gemm_nt %-40s %8.1f us  %6.1f TFLOP/s" % (name, dt * 1e6, fl / dt / 1e12))

# ---- depthwise
H = 512
for d in (1, 8, 128):
    x, y = act(H), act(H)
    w, bias = vec(H * 3).view(H, 3), vec(H)
    st = ops.Stat(vec(B), vec(B).abs() + 0.5, 1, 0, False)
    st.c1, st.c2 = vec(B), vec(B)
    alpha = torch.tensor([0.25], device=dev)
    g, be = vec(H), vec(H)
    acc = ops.Accum(B, T, False, dev)
    dt = timeit(lambda: ops.dw_fwd(x, w, bias, y, d, d, xflags=3, alpha_in=alpha, stat=st, gamma=g, beta=be,
                                   alpha_out=alpha, acc=acc))
    by = 2.0 * B * H * T * 4
    print("dw_fwd d=%-3d %8.1f us  %7.1f GB/s (algorithmic 2 E_H)" % (d, dt * 1e6, by / dt / 1e9))
    dn2, dx = act(H), act(H)
    dw, db = torch.zeros(H, 3, device=dev), torch.zeros(H, device=dev)
    dg1, db1, dg2, db2 = (torch.zeros(H, device=dev) for _ in range(4))
    da2 = torch.zeros(32, device=dev, dtype=torch.float64)
    iacc = ops.Accum(B, T, False, dev)
    dt = timeit(lambda: ops.dw_bwd(dn2, x, w, dw, db, dx, d, d, xflags=3, alpha1=alpha, stat1=st, gamma1=g, beta1=be,
                                   dgamma1=dg1, dbeta1=db1, iacc=iacc, y2=y, alpha2=alpha, stat2=st, gamma2=g,
                                   dgamma2=dg2, dbeta2=db2, dalpha2=da2))
    by = 4.0 * B * H * T * 4
    print("dw_bwd d=%-3d %8.1f us  %7.1f GB/s (algorithmic 4 E_H)" % (d, dt * 1e6, by / dt / 1e9))

# ---- codec
N, L, S = 512, 16, 8
w = act(N); mask = act(2 * N).uniform_(0.1, 0.9)
Wd = vec(N * L).view(N, 1, L)
out = torch.empty(B, 2, 32000, device=dev)
dt = timeit(lambda: ops.maskdecode_fwd(w, mask, 2, Wd, S, out, 0))
print("maskdecode_fwd %8.1f us  %7.1f GB/s (3 E_N)" % (dt * 1e6, 3.0 * B * N * T * 4 / dt / 1e9))
dz, dwd = act(2 * N), act(N)
dWd = torch.zeros(N, L, device=dev)
dout = torch.randn(B, 2, 32000, device=dev)
dt = timeit(lambda: ops.maskdecode_bwd(dout, 0, w, mask, 2, Wd, S, True, dz, dwd, dWd))
print("maskdecode_bwd %8.1f us  %7.1f GB/s (6 E_N)" % (dt * 1e6, 6.0 * B * N * T * 4 / dt / 1e9))
x2 = torch.randn(B, 32000, device=dev)
dn = act(N)
st = ops.Stat(vec(B), vec(B).abs() + 0.5, 1, 0, False); st.c1, st.c2 = vec(B), vec(B)
dW = torch.zeros(N, L, device=dev); dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
dt = timeit(lambda: ops.encoder_bwd(x2, 0, N, L, S, False, w, dW, dw_a=dwd, dn=dn, stat=st, gamma=vec(N), dgamma=dg, dbeta=db))
print("encoder_bwd    %8.1f us  %7.1f GB/s (3 E_N)" % (dt * 1e6, 3.0 * B * N * T * 4 / dt / 1e9))
