"""Tensor-level wrappers over the C ABI of libctn_hip.so.

Nothing here computes on the host or falls back to PyTorch kernels: each
function checks layouts, fills the argument struct and enqueues HIP kernels on
the current torch stream.  Activations are (B, C, T) fp32 views with unit time
stride whose row and batch strides are multiples of 4 floats ("act" layout,
see new_act); anything else is copied into that layout first.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (XF_PRELU, XF_NORM, XF_NORMBWD, EPI_STORE, EPI_STATS, EPI_HEADS, EPI_SIGMOID,  # noqa: F401
                   EPI_PRELUBWD, EPI_NORMBWD, EPI_ADD)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    if not torch.cuda.is_available():
        raise _lib.CtnError("speech_amd operators need an MI355X (no GPU visible); there is no CPU path")
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CtnError("speech_amd operators run on the GPU only (got a {} tensor); "
                                "there is no CPU path".format(t.device))


def round4(n):
    return (n + 3) // 4 * 4


def new_act(B, Cc, T, device, zero=False):
    """(B, C, T) view of a (B, C, ld) buffer, ld = T rounded up to 4 floats (16-byte rows)."""
    ld = round4(T)
    buf = (torch.zeros if zero else torch.empty)((B, Cc, ld), device=device, dtype=torch.float32)
    return buf[:, :, :T]


def is_act(x):
    return (x.dim() == 3 and x.dtype == torch.float32 and x.is_cuda and x.stride(2) == 1
            and x.stride(1) % 4 == 0 and x.stride(0) % 4 == 0 and x.stride(1) >= x.shape[2]
            and x.data_ptr() % 16 == 0)


def as_act(x):
    if is_act(x):
        return x
    _require_cuda(x)
    y = new_act(x.shape[0], x.shape[1], x.shape[2], x.device)
    y.copy_(x)
    return y


class Stat:
    """Layer-norm statistics: mean/rstd (and backward c1/c2) indexed [b*sb + t*st]."""

    __slots__ = ("mean", "rstd", "c1", "c2", "sb", "st", "cln")

    def __init__(self, mean, rstd, sb, st, cln):
        self.mean, self.rstd, self.sb, self.st, self.cln = mean, rstd, sb, st, cln
        self.c1 = self.c2 = None


class ZeroArena:
    """One zero-filled allocation handed out in 256-byte aligned slices: a forward or backward pass needs about a
    hundred small zeroed accumulators, and a fill kernel per accumulator costs more than the accumulators' users
    save.  A slice keeps the whole allocation alive; when the arena is used up, take() returns None."""

    def __init__(self, device, nbytes):
        self.buf = torch.zeros((nbytes + 7) // 8, device=device, dtype=torch.float64)
        self.device, self.off, self.nbytes = self.buf.device, 0, self.buf.numel() * 8

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= d
        item = 8 if dtype == torch.float64 else 4
        nbytes = (n * item + 255) // 256 * 256
        if self.off + nbytes > self.nbytes:
            return None
        lo = self.off // 8
        self.off += nbytes
        flat = self.buf[lo:lo + nbytes // 8]
        if dtype != torch.float64:
            flat = flat.view(dtype)
        return flat[:n].view(shape)


_arena = None


class zero_arena:
    """with zero_arena(device, nbytes): zeroed accumulators created inside come from one allocation."""

    def __init__(self, device, nbytes):
        self.device, self.nbytes = device, nbytes

    def __enter__(self):
        global _arena
        self.prev = _arena
        _arena = ZeroArena(self.device, self.nbytes)
        return _arena

    def __exit__(self, *exc):
        global _arena
        _arena = self.prev
        return False


def zeros(shape, device, dtype):
    """Zeroed tensor, from the active arena when there is one on this device with room left."""
    if _arena is not None and _arena.device == torch.device(device):
        t = _arena.take(tuple(shape), dtype)
        if t is not None:
            return t
    return torch.zeros(shape, device=device, dtype=dtype)


def accum_bytes(B, T, cln):
    """Arena bytes of one Accum (see ZeroArena.take's rounding)."""
    n = B * 2 * round4(T) * 4 if cln else B * _lib.STAT_SLOTS * 2 * 8
    extra = 0 if cln else (B * 4 + 255) // 256 * 256              # the gLN accumulator's ticket counters
    return (n + 255) // 256 * 256 + extra


class Accum:
    """Zeroed accumulators for one norm's statistics: double (B, STAT_SLOTS, 2) for gLN (slot-replicated, the
    value is the sum over slots), float (B, 2, ld) for cLN."""

    __slots__ = ("sums", "col", "ld", "count")

    def __init__(self, B, T, cln, device):
        if cln:
            self.ld = round4(T)
            self.col = zeros((B, 2, self.ld), device, torch.float32)
            self.sums = self.count = None
        else:
            self.ld = 0
            self.col = None
            self.sums = zeros((B, _lib.STAT_SLOTS, 2), device, torch.float64)
            self.count = zeros((B,), device, torch.int32)         # one ticket per producing block (ctn_finish)


# A producer of gLN sums can finish the reduction itself (ctn_finish in include/ctn.h): the block that adds the last
# partial sums writes mean / rstd (or the backward coefficients), and the stand-alone finalize launch disappears.
# Measured, it does not pay at either end: a block must see its additions performed before it takes its ticket, and that
# wait also covers the block's output stores, which it otherwise leaves in flight when it retires.  The depthwise
# kernels, whose blocks stream one row and retire, run at less than half their rate that way (85 -> 202 us per launch at
# batch 32); the GEMM epilogues lose 2 %, which at batch 32 cancels the 49 launches saved and at batch 2 costs 3 % of the
# step.  Hence off by default; CTN_FOLD_FINISH = all | gemm | 0 chooses the producers that fold.
def _initial_fold():
    v = os.environ.get("CTN_FOLD_FINISH", "0").strip().lower()
    return {"all": ("gemm", "dw"), "1": ("gemm", "dw"), "gemm": ("gemm",), "0": (), "none": ()}.get(v, ())


FOLD_FINISH = _initial_fold()         # producers that finish their reductions: any of "gemm", "dw"


def stats_request(acc, B, Cc, T, eps, device, producer="gemm"):
    """(stat, fin) for a producer ("gemm" or "dw") about to fill `acc` with forward statistics: fin goes to the
    producer's `fin=` argument and stat is complete when the producer has run; (None, None) when the reduction is not
    finished in the producer (cLN, or not switched on for this kind of producer) -- call finalize_stats() behind it then."""
    if producer not in FOLD_FINISH or acc.sums is None:
        return None, None
    stat = Stat(torch.empty((B,), device=device, dtype=torch.float32),
                torch.empty((B,), device=device, dtype=torch.float32), 1, 0, False)
    return stat, (_lib.FIN_STATS, acc, stat.mean, stat.rstd, None, float(Cc) * float(T), eps)


def coefs_request(acc, stat, B, Cc, T, producer="gemm"):
    """fin for a producer about to fill `acc` with the backward sums of `stat`'s norm: stat.c1 / stat.c2 are complete
    when the producer has run.  None when the producer does not finish them: call bwd_coefs() behind it then."""
    if producer not in FOLD_FINISH or acc.sums is None or stat.cln:
        return None
    stat.c1 = torch.empty_like(stat.mean)
    stat.c2 = torch.empty_like(stat.mean)
    return (_lib.FIN_COEFS, acc, stat.c1, stat.c2, stat.rstd, float(Cc) * float(T), 0.0)


def _set_fin(fin_struct, fin):
    if fin is None:
        return
    kind, acc, out0, out1, rstd, n, eps = fin
    fin_struct.count, fin_struct.out0, fin_struct.out1 = _ptr(acc.count), _ptr(out0), _ptr(out1)
    fin_struct.rstd, fin_struct.n, fin_struct.eps, fin_struct.kind = _ptr(rstd), n, eps, kind


# ------------------------------------------------------------------------------------------------ arithmetic mode
# The library keeps no mode: every GEMM call carries its arithmetic (CTN_ARITH_* in include/ctn.h) and tile width in its
# argument struct.  What this module passes by default lives HERE, on the caller's side of the C ABI:
#   pieces 3 (the default): fp32 operands split exactly into three bf16 pieces, six products on the bf16 MFMA, fp32
#       accumulation -- the f32-MFMA bounds hold in every test;
#   pieces 0: f32 MFMAs;  pieces 2: two pieces, three products (each GEMM within 5e-5 instead of 2e-5 of exact math).
# CTN_GEMM_SPLIT=<pieces> in the environment sets the initial value.
def _initial_split():
    v = os.environ.get("CTN_GEMM_SPLIT", "3").strip()
    return {"0": 0, "1": 2, "2": 2, "3": 3}.get(v, 3)


_gemm_split = _initial_split()
_gemm_tile = 0
_prep_kernel = int(os.environ.get("CTN_PREP_KERNEL", "0"))      # CTN_PREP_* for calls that do not say (diagnostics)


def set_gemm_split(pieces):
    """Arithmetic of the GEMM calls made through this module from now on: 0 / False = f32 MFMAs, 2 / True = two bf16
    pieces and three products, 3 = three pieces and six products (all 24 mantissa bits; the default).  HIP graphs captured
    earlier keep the calls they captured."""
    global _gemm_split
    level = 2 if pieces is True else int(pieces)
    if level not in (0, 2, 3):
        raise ValueError("pieces is 0 (f32 MFMA), 2 or 3")
    _gemm_split = level


def get_gemm_split():
    """0 (f32 MFMAs), 2 or 3: the number of bf16 pieces per operand the GEMM calls of this module use."""
    return _gemm_split


def set_gemm_tile(width):
    """Forces the output-tile width (64 or 128 columns) of the pointwise GEMM's LDS-DMA kernel for calls made through this
    module; 0 = let the library choose per launch (the default).  A tuning and testing aid."""
    global _gemm_tile
    if width not in (0, 64, 128):
        raise ValueError("width is 0, 64 or 128")
    _gemm_tile = width


def _arith():
    return {0: _lib.ARITH_F32, 2: _lib.ARITH_BF16X2, 3: _lib.ARITH_BF16X3}[_gemm_split]


# ------------------------------------------------------------------------------------------------ chunking
def segment(x, K, P, S, pad_left=0):
    """(B, C, T) -> dense (B, C, S, K): chunk s holds frames s*P .. s*P+K-1 of x zero-padded by pad_left in front."""
    lib = _lib.load()
    _require_cuda(x)
    B, Cc, T = x.shape
    assert x.stride(2) == 1
    y = torch.empty((B, Cc, S, K), device=x.device, dtype=torch.float32)
    _lib.check(lib.ctn_segment(_ptr(x), x.stride(0), x.stride(1), B, Cc, T, pad_left, K, P, S, _ptr(y), _stream()),
               "ctn_segment")
    return y


def overlap_add(y, P, T, pad_left=0):
    """dense (B, C, S, K) -> activation (B, C, T): frame t sums the chunk entries that cover padded position t + pad_left."""
    lib = _lib.load()
    _require_cuda(y)
    B, Cc, S, K = y.shape
    assert y.is_contiguous()
    x = new_act(B, Cc, T, y.device)
    _lib.check(lib.ctn_overlap_add(_ptr(y), B, Cc, S, K, P, pad_left, T, _ptr(x), x.stride(0), x.stride(1), _stream()),
               "ctn_overlap_add")
    return x


# ------------------------------------------------------------------------------------------------ recurrence
LSTM_MAX_HIDDEN = 128


def lstm_fwd(xproj, whh):
    """xproj (N, T, D, 4H) dense fp32, whh (D, 4H, H) -> hout (N, T, D, H), save (N, T, D, 5, H); see ctn_lstm_fwd."""
    lib = _lib.load()
    _require_cuda(xproj, whh)
    N, T, D, G = xproj.shape
    H = G // 4
    assert xproj.is_contiguous() and whh.is_contiguous() and tuple(whh.shape) == (D, 4 * H, H)
    hout = torch.empty((N, T, D, H), device=xproj.device, dtype=torch.float32)
    save = torch.empty((N, T, D, 5, H), device=xproj.device, dtype=torch.float32)
    _lib.check(lib.ctn_lstm_fwd(_ptr(xproj), _ptr(whh), N, T, H, D, _ptr(hout), _ptr(save), _stream()), "ctn_lstm_fwd")
    return hout, save


def lstm_bwd(dhout, save, whh, bias_grad=True):
    """dhout (N, T, D, H) -> dgates (N, T, D, 4H) = gradient w.r.t. xproj, and its sum over (N, T) (D, 4H) = the bias
    gradient (None if not asked for); see ctn_lstm_bwd."""
    lib = _lib.load()
    _require_cuda(dhout, save, whh)
    N, T, D, H = dhout.shape
    assert dhout.is_contiguous() and save.is_contiguous() and tuple(save.shape) == (N, T, D, 5, H)
    assert whh.is_contiguous() and tuple(whh.shape) == (D, 4 * H, H)
    dgates = torch.empty((N, T, D, 4 * H), device=dhout.device, dtype=torch.float32)
    dbias = torch.empty((D, 4 * H), device=dhout.device, dtype=torch.float32) if bias_grad else None
    _lib.check(lib.ctn_lstm_bwd(_ptr(dhout), _ptr(save), _ptr(whh), N, T, H, D, _ptr(dgates),
                                _ptr(dbias) if bias_grad else None, _stream()), "ctn_lstm_bwd")
    return dgates, dbias


# ------------------------------------------------------------------------------------------------ products over rows
def _rows2d(t):
    """A (rows, features) fp32 view with unit feature stride (row stride arbitrary), or a contiguous copy of it."""
    _require_cuda(t)
    assert t.dim() == 2 and t.dtype == torch.float32
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def rows_gemm(x, w, bias=None, w_is_km=False, arith=None):
    """x (R, K) rows -> y (R, M) = x w^T + bias for w (M, K) (w_is_km False: nn.Linear's layout), or x w for w (K, M)
    (w_is_km True: the input gradient of such a layer, the weight as it lies); see ctn_rows_gemm.  arith: CTN_ARITH_*
    (None: this module's current GEMM arithmetic, set_gemm_split).  In the three-piece arithmetic a (K, M) weight -- a
    small matrix -- is transposed first (ctn_transpose), so that the product takes the bf16 kernel."""
    lib = _lib.load()
    x, w = _rows2d(x), _rows2d(w)
    arith = (_lib.ARITH_F32 if _gemm_split == 0 else _lib.ARITH_BF16X3) if arith is None else arith
    if w_is_km and arith != _lib.ARITH_F32:
        wt = torch.empty((w.shape[1], w.shape[0]), device=w.device, dtype=torch.float32)
        transpose(w.contiguous(), wt, w.shape[0])
        w, w_is_km = wt, False
    R, K = x.shape
    M = w.shape[1] if w_is_km else w.shape[0]
    assert (w.shape[0] if w_is_km else w.shape[1]) == K
    y = torch.empty((R, M), device=x.device, dtype=torch.float32)
    if R == 0:
        return y
    if bias is not None:
        _require_cuda(bias)
        assert bias.dtype == torch.float32 and bias.numel() == M and bias.is_contiguous()
    _lib.check(lib.ctn_rows_gemm(_ptr(x), _ld(x), _ptr(w), _ld(w), 1 if w_is_km else 0, _ptr(bias), R, K, M, _ptr(y), M,
                                 arith, _stream()), "ctn_rows_gemm")
    return y


def rows_wgrad(dy, x, dW=None, dbias=None, sign=1.0, arith=None):
    """dW (M, K) += sign * dy^T x over the rows of dy (R, M) and x (R, K); dbias (M,) += sign * column sums of dy when
    given.  dW None: a zeroed (M, K) tensor is made and returned; see ctn_rows_wgrad."""
    lib = _lib.load()
    dy, x = _rows2d(dy), _rows2d(x)
    R, M = dy.shape
    K = x.shape[1]
    assert x.shape[0] == R
    if dW is None:
        dW = torch.zeros((M, K), device=dy.device, dtype=torch.float32)
    assert dW.dtype == torch.float32 and tuple(dW.shape) == (M, K) and (K == 1 or dW.stride(1) == 1)
    if dbias is not None:
        assert dbias.dtype == torch.float32 and dbias.numel() == M and dbias.is_contiguous()
    if R == 0:
        return dW
    arith = _lib.ARITH_DEFAULT if arith is None else arith        # default: the library picks the faster form per shape
    _lib.check(lib.ctn_rows_wgrad(_ptr(dy), _ld(dy), _ptr(x), _ld(x), R, M, K, _ptr(dW), _ld(dW), _ptr(dbias), float(sign),
                                  arith, _stream()), "ctn_rows_wgrad")
    return dW


# ------------------------------------------------------------------------------------------------ axis shuffles
def _reaches(t, last):
    """True if element offset `last` (from t's first element) lies inside t's storage."""
    return t.storage_offset() + last < t.untyped_storage().nbytes() // 4


def permute(src, dst, M, K, lds, ldd, nb1=1, sb1=0, db1=0, nb2=1, sb2=0, db2=0, addend=None):
    """dst[b2][b1][k][m] = src[b2][b1][m][k] (+ addend, laid out like dst); strides in elements from the tensors' first
    elements; see ctn_permute."""
    lib = _lib.load()
    _require_cuda(src, dst, addend)
    assert src.dtype == torch.float32 and dst.dtype == torch.float32
    assert addend is None or (addend.dtype == torch.float32 and addend.stride() == dst.stride() and addend.shape == dst.shape)
    last_src = (nb2 - 1) * sb2 + (nb1 - 1) * sb1 + (M - 1) * lds + K - 1
    last_dst = (nb2 - 1) * db2 + (nb1 - 1) * db1 + (K - 1) * ldd + M - 1
    assert _reaches(src, last_src) and _reaches(dst, last_dst), "permute: strides reach outside the tensors"
    _lib.check(lib.ctn_permute(_ptr(src), _ptr(dst), _ptr(addend) if addend is not None else None, M, K, lds, ldd, nb1, sb1,
                               db1, nb2, sb2, db2, _stream()), "ctn_permute")
    return dst


def rows_from_blocks(x, walk_chunks):
    """(B, F, S, K) contiguous -> (B, S, K, F) (walk_chunks False) or (B, K, S, F) (True): the recurrences' rows."""
    B, F_, S, K = x.shape
    assert x.is_contiguous()
    y = torch.empty((B, K, S, F_) if walk_chunks else (B, S, K, F_), device=x.device, dtype=torch.float32)
    if walk_chunks:         # per (b, s): (F, K) -> (K, F)
        return permute(x, y, F_, K, S * K, S * F_, nb1=S, sb1=K, db1=F_, nb2=B, sb2=F_ * S * K, db2=F_ * S * K)
    return permute(x, y, F_, S * K, S * K, F_, nb1=B, sb1=F_ * S * K, db1=F_ * S * K)


def blocks_from_rows(r, walk_chunks):
    """Inverse of rows_from_blocks: (B, S, K, F) or (B, K, S, F) contiguous -> (B, F, S, K)."""
    assert r.is_contiguous()
    if walk_chunks:
        B, K, S, F_ = r.shape
        x = torch.empty((B, F_, S, K), device=r.device, dtype=torch.float32)
        return permute(r, x, K, F_, S * F_, S * K, nb1=S, sb1=F_, db1=K, nb2=B, sb2=F_ * S * K, db2=F_ * S * K)
    B, S, K, F_ = r.shape
    x = torch.empty((B, F_, S, K), device=r.device, dtype=torch.float32)
    return permute(r, x, S * K, F_, F_, S * K, nb1=B, sb1=F_ * S * K, db1=F_ * S * K)


def feature_first(r):
    """(B, L, F) contiguous -> (B, F, L) in the activation layout (16-byte rows) the norm kernels read."""
    B, L, F_ = r.shape
    assert r.is_contiguous()
    y = new_act(B, F_, L, r.device)
    return permute(r, y, L, F_, F_, y.stride(1), nb1=B, sb1=L * F_, db1=y.stride(0))


def feature_last(g):
    """(B, F, L) with unit stride along L -> (B, L, F) contiguous."""
    B, F_, L = g.shape
    assert g.stride(2) == 1
    y = torch.empty((B, L, F_), device=g.device, dtype=torch.float32)
    return permute(g, y, F_, L, g.stride(1), F_, nb1=B, sb1=g.stride(0), db1=L * F_)


def swap_blocks(z, addend=None, act=False):
    """(B, F, P, Q) with contiguous (P, Q) planes -> (B, F, Q, P) (+ addend of that shape).  act=True lays the result
    out as a (B, F, Q*P) activation (16-byte rows) and returns its (B, F, Q, P) view."""
    B, F_, P, Q = z.shape
    assert z.stride(3) == 1 and z.stride(2) == Q and B * F_ <= 65535
    if act:
        flat = new_act(B, F_, Q * P, z.device)
        y = flat.view(B, F_, Q, P)
    else:
        y = torch.empty((B, F_, Q, P), device=z.device, dtype=torch.float32)
    return permute(z, y, P, Q, Q, P, nb1=F_, sb1=z.stride(1), db1=y.stride(1), nb2=B, sb2=z.stride(0), db2=y.stride(0),
                   addend=addend)


# ------------------------------------------------------------------------------------------------ norms
def transpose(src2d, dst, ldd, coff=0):
    lib = _lib.load()
    M, K = src2d.shape
    _lib.check(lib.ctn_transpose(_ptr(src2d), M, K, _ptr(dst), ldd, coff, _stream()), "ctn_transpose")


def transpose_items(entries):
    """ctypes array of ctn_transpose_item from (src2d (M, K) contiguous fp32, dst, ldd, coff) tuples; the caller keeps the
    tensors alive."""
    arr = (_lib.TransposeItem * len(entries))()
    for it, (src, dst, ldd, coff) in zip(arr, entries):
        _require_cuda(src, dst)
        assert src.dim() == 2 and src.is_contiguous() and src.dtype == torch.float32 and dst.dtype == torch.float32
        assert dst.is_contiguous() and dst.numel() >= (src.shape[1] - 1) * ldd + coff + src.shape[0]
        it.src, it.dst, it.ldd, it.coff, it.M, it.K = _ptr(src), _ptr(dst), ldd, coff, src.shape[0], src.shape[1]
    return arr


def transpose_batch(items):
    lib = _lib.load()
    _lib.check(lib.ctn_transpose_batch(items, len(items), _stream()), "ctn_transpose_batch")


def prep_usable(M, K, K1=None):
    """True when ctn_gemm_nn takes a prepared A for this shape in the current arithmetic (three bf16 pieces)."""
    K1 = K if K1 is None else K1
    return _gemm_split == 3 and M > 64 and M % 4 == 0 and K % 16 == 0 and K1 % 16 == 0 and K <= 4096


def weight_planes_numel(K, M):
    """bf16 elements of a prepared K x M matrix (three planes, K padded to 16, M to 128)."""
    return 3 * ((K + 15) // 16 * 16) * ((M + 127) // 128 * 128)


def wprep_items(entries):
    """ctypes array of ctn_wprep_item from (src2d, transpose, k_off, m_off, K, M, dst) tuples; src2d a (rows, cols) fp32
    view with unit column stride, dst the prepared matrix (bf16 tensor of weight_planes_numel(K, M) elements, zero
    where no item writes).  The array holds raw addresses: the caller keeps the tensors alive."""
    arr = (_lib.WPrepItem * len(entries))()
    for it, (src, transpose, k_off, m_off, K, M, dst) in zip(arr, entries):
        _require_cuda(src, dst)
        assert src.dim() == 2 and src.dtype == torch.float32 and src.stride(1) == 1
        assert dst.dtype == torch.bfloat16 and dst.is_contiguous() and dst.numel() >= weight_planes_numel(K, M)
        it.src, it.ld, it.dst = _ptr(src), src.stride(0), _ptr(dst)
        it.rows, it.cols, it.transpose = src.shape[0], src.shape[1], 1 if transpose else 0
        it.k_off, it.m_off, it.K, it.M = k_off, m_off, K, M
    return arr


def weight_prep(items):
    """Runs ctn_weight_prep over a wprep_items() array on the current stream."""
    lib = _lib.load()
    _lib.check(lib.ctn_weight_prep(items, len(items), _stream()), "ctn_weight_prep")


def prepared(K, M, parts, device):
    """A freshly prepared K x M matrix from parts = [(src2d, transpose, k_off, m_off), ...]."""
    dst = torch.zeros(weight_planes_numel(K, M), device=device, dtype=torch.bfloat16)
    weight_prep(wprep_items([(src, tr, ko, mo, K, M, dst) for src, tr, ko, mo in parts]))
    return dst


def norm_stats(x, alpha, acc):
    lib = _lib.load()
    B, Cc, T = x.shape
    _lib.check(lib.ctn_norm_stats(_ptr(x), B, Cc, T, x.stride(1), _ptr(alpha), _ptr(acc.sums), _ptr(acc.col),
                                  acc.ld, _stream()), "ctn_norm_stats")


def finalize_stats(acc, B, Cc, T, eps, device):
    """Accumulated sums -> Stat (mean, rstd)."""
    lib = _lib.load()
    if acc.col is not None:
        ld = acc.ld
        mean = torch.empty((B, ld), device=device, dtype=torch.float32)
        rstd = torch.empty((B, ld), device=device, dtype=torch.float32)
        _lib.check(lib.ctn_cln_finalize(_ptr(acc.col), B, Cc, T, ld, eps, _ptr(mean), _ptr(rstd), ld, _stream()),
                   "ctn_cln_finalize")
        return Stat(mean, rstd, ld, 1, True)
    mean = torch.empty((B,), device=device, dtype=torch.float32)
    rstd = torch.empty((B,), device=device, dtype=torch.float32)
    _lib.check(lib.ctn_gln_finalize(_ptr(acc.sums), B, float(Cc) * float(T), eps, _ptr(mean), _ptr(rstd), _stream()),
               "ctn_gln_finalize")
    return Stat(mean, rstd, 1, 0, False)


def bwd_coefs(acc, stat, B, Cc, T, eps):
    """Backward sums -> stat.c1, stat.c2."""
    lib = _lib.load()
    stat.c1 = torch.empty_like(stat.mean)
    stat.c2 = torch.empty_like(stat.mean)
    if stat.cln:
        _lib.check(lib.ctn_cln_bwd_coefs(_ptr(acc.col), acc.ld, _ptr(stat.mean), _ptr(stat.rstd), stat.sb, B, Cc, T,
                                         eps, _ptr(stat.c1), _ptr(stat.c2), _stream()), "ctn_cln_bwd_coefs")
    else:
        _lib.check(lib.ctn_gln_bwd_coefs(_ptr(acc.sums), _ptr(stat.rstd), B, float(Cc) * float(T), _ptr(stat.c1),
                                         _ptr(stat.c2), _stream()), "ctn_gln_bwd_coefs")
    return stat


def heads_coefs_usable(M, M1, H):
    """True when ctn_gemm_nt can deliver the by-products heads_bwd_coefs needs for these head shapes."""
    return M % 16 == 0 and M1 % 16 == 0 and H % 16 == 0 and M >= 16 and H >= 16


def heads_bwd_coefs(W, W2, gamma, beta, rowsum_b, wdot_sums, stat, B, H, T, dW, dW2):
    """stat.c1, stat.c2 of the gLN in front of the heads from the weight-gradient launch's by-products, and the beta
    term of the heads' weight gradient (see ctn_heads_bwd_coefs in include/ctn.h).  W / dW: (M1, H) views or None,
    W2 / dW2: (M - M1, H) views or None."""
    lib = _lib.load()
    M1 = 0 if W is None else W.shape[0]
    M = M1 + (0 if W2 is None else W2.shape[0])
    stat.c1 = torch.empty_like(stat.mean)
    stat.c2 = torch.empty_like(stat.mean)
    work = zeros((B + 1,), rowsum_b.device, torch.float64)            # S1 accumulators and, in the last slot, the ticket
    _lib.check(lib.ctn_heads_bwd_coefs(_ptr(W), 0 if W is None else W.stride(0), _ptr(W2), 0 if W2 is None else W2.stride(0),
                                       M, M1, H, _ptr(gamma), _ptr(beta), _ptr(rowsum_b), _ptr(wdot_sums), _ptr(stat.rstd),
                                       B, float(H) * float(T), _ptr(stat.c1), _ptr(stat.c2), _ptr(dW),
                                       0 if dW is None else dW.stride(0), _ptr(dW2), 0 if dW2 is None else dW2.stride(0),
                                       work.data_ptr(), work.data_ptr() + 8 * B, _stream()), "ctn_heads_bwd_coefs")
    return stat


def norm_apply(x, alpha, stat, gamma, beta, y):
    lib = _lib.load()
    B, Cc, T = x.shape
    _lib.check(lib.ctn_norm_apply(_ptr(x), B, Cc, T, x.stride(1), _ptr(alpha), _ptr(stat.mean), _ptr(stat.rstd),
                                  stat.sb, stat.st, _ptr(gamma), _ptr(beta), _ptr(y), y.stride(1), _stream()),
               "ctn_norm_apply")


def norm_bwd_reduce(dn, x, alpha, stat, gamma, acc, dgamma, dbeta):
    lib = _lib.load()
    B, Cc, T = x.shape
    _lib.check(lib.ctn_norm_bwd_reduce(_ptr(dn), dn.stride(1), _ptr(x), x.stride(1), B, Cc, T, _ptr(alpha),
                                       _ptr(stat.mean), _ptr(stat.rstd), stat.sb, stat.st, _ptr(gamma),
                                       _ptr(acc.sums), _ptr(acc.col), acc.ld, _ptr(dgamma), _ptr(dbeta), _stream()),
               "ctn_norm_bwd_reduce")


def norm_bwd_apply(dn, x, alpha, stat, gamma, dx, dalpha):
    lib = _lib.load()
    B, Cc, T = x.shape
    _lib.check(lib.ctn_norm_bwd_apply(_ptr(dn), dn.stride(1), _ptr(x), x.stride(1), B, Cc, T, _ptr(alpha),
                                      _ptr(stat.mean), _ptr(stat.rstd), _ptr(stat.c1), _ptr(stat.c2), stat.sb,
                                      stat.st, _ptr(gamma), _ptr(dx), dx.stride(1), _ptr(dalpha), _stream()),
               "ctn_norm_bwd_apply")


# ------------------------------------------------------------------------------------------------ GEMMs
def gemm_nn(M, T, K, batch, A, lda, X, Y, *, A2=None, lda2=0, K1=None, Xb=None, xflags=0, X2=None, alpha=None,
            stat=None, gamma=None, beta=None, bias=None, bias2=None, emode=EPI_STORE, Mo=0, R=None, Y2=None,
            accumulate=False, ealpha=None, eacc=None, E=None, estat=None, egamma=None, esums=None, tile_width=0,
            A_prep=None, fin=None, prep_kernel=0):
    """Y[b] = A(k-major) . xform(X[b]); see ctn_gemm_nn in include/ctn.h.  X/Xb/X2/Y/Y2/R/E are act views.  A_prep: the
    same A as a prepared matrix (prepared() / weight_prep()); A, A2 may then be None.  prep_kernel: CTN_PREP_* (0 = the
    library's choice, 1 = wave-split, 2 = block-split, 3 = eight-wave wave-split kernel for the prepared operand)."""
    lib = _lib.load()
    a = _lib.GemmNNArgs()
    a.tile_width = tile_width or _gemm_tile
    a.arith = _arith()
    a.M, a.T, a.K, a.batch = M, T, K, batch
    a.A, a.lda = _ptr(A), lda
    a.A2, a.lda2 = _ptr(A2), lda2
    a.K1 = K if K1 is None else K1
    a.xflags = xflags
    a.bias, a.bias2 = _ptr(bias), _ptr(bias2)
    a.X, a.ldx, a.xbs = _ptr(X), X.stride(1), X.stride(0)
    if Xb is not None:
        a.Xb, a.ldxb, a.xbbs = _ptr(Xb), Xb.stride(1), Xb.stride(0)
    if X2 is not None:
        a.X2, a.ldx2, a.x2bs = _ptr(X2), X2.stride(1), X2.stride(0)
    a.alpha = _ptr(alpha)
    if stat is not None:
        a.mean, a.rstd, a.c1, a.c2 = _ptr(stat.mean), _ptr(stat.rstd), _ptr(stat.c1), _ptr(stat.c2)
        a.ssb, a.sst = stat.sb, stat.st
    a.gamma, a.beta = _ptr(gamma), _ptr(beta)
    a.Y, a.ldy, a.ybs = _ptr(Y), Y.stride(1), Y.stride(0)
    a.emode, a.Mo = emode, Mo
    if R is not None:
        a.R, a.ldr, a.rbs = _ptr(R), R.stride(1), R.stride(0)
    if Y2 is not None:
        a.Y2, a.ldy2, a.y2bs = _ptr(Y2), Y2.stride(1), Y2.stride(0)
    a.accumulate = 1 if accumulate else 0
    a.ealpha = _ptr(ealpha)
    if eacc is not None:
        a.esums = _ptr(eacc.sums)
        a.ecol, a.ecol_bs, a.ecol_ld = _ptr(eacc.col), 2 * eacc.ld, eacc.ld
    if esums is not None:
        a.esums = _ptr(esums)
    if E is not None:
        a.E, a.lde, a.ebs = _ptr(E), E.stride(1), E.stride(0)
    if estat is not None:
        a.emean, a.erstd, a.essb, a.esst = _ptr(estat.mean), _ptr(estat.rstd), estat.sb, estat.st
    a.egamma = _ptr(egamma)
    a.A_prep = _ptr(A_prep)
    a.prep_kernel = prep_kernel or _prep_kernel
    _set_fin(a.fin, fin)
    _lib.check(lib.ctn_gemm_nn(C.byref(a), _stream()), "ctn_gemm_nn")


def pick_kchunk(T, batch, tiles, target_blocks=1024):
    """Time steps per wgrad block: about one generation of resident blocks on the 256 CUs (every block ends with atomic
    adds of its whole output tile, the cost that grows with the number of chunks), chunks >= 256 frames.  The number of
    chunks is ROUNDED to the nearest: 512 blocks per chunk against 768 slots run faster as 1024 blocks (a third of a
    generation of tail) than as 512 (a third of the chip's slots empty throughout): 288 -> 265 us on the heads' launch."""
    chunks = max(1, int(target_blocks / max(1, batch * tiles) + 0.5))
    k = (T + chunks - 1) // chunks
    k = max(256, (k + 31) // 32 * 32)
    return k


def gemm_nt(M, N, T, batch, A, B, Cout, ldc, *, Ab=None, M1=None, C2=None, ldc2=0, aflags=0, A2=None,
            a_alpha=None, a_stat=None, a_gamma=None, a_rowsum=None, a_rowsum2=None, a_dalpha=None, bflags=0,
            b_alpha=None, b_stat=None, b_gamma=None, b_beta=None, kchunk=None, a_rowsum_b=None, wdot=None, wdot2=None,
            wdot_sums=None):
    """C[m][n] += sum_b,t xformA(A)[b][m][t] * xformB(B)[b][n][t]; see ctn_gemm_nt in include/ctn.h.  a_rowsum_b
    (batch, M) float and wdot / wdot2 (2-D views in C's / C2's layout) with wdot_sums (batch, STAT_SLOTS) double: the
    per-batch by-products ctn_heads_bwd_coefs consumes, all zeroed by the caller."""
    lib = _lib.load()
    a = _lib.GemmNTArgs()
    a.arith = _arith()
    a.M, a.N, a.T, a.batch = M, N, T, batch
    a.M1 = M if M1 is None else M1
    if A is not None:
        a.A, a.lda, a.abs_ = _ptr(A), A.stride(1), A.stride(0)
    if Ab is not None:
        a.Ab, a.ldab, a.abbs = _ptr(Ab), Ab.stride(1), Ab.stride(0)
    a.aflags = aflags
    if A2 is not None:
        a.A2, a.lda2, a.a2bs = _ptr(A2), A2.stride(1), A2.stride(0)
    a.a_alpha = _ptr(a_alpha)
    if a_stat is not None:
        a.a_mean, a.a_rstd, a.a_c1, a.a_c2 = _ptr(a_stat.mean), _ptr(a_stat.rstd), _ptr(a_stat.c1), _ptr(a_stat.c2)
        a.a_ssb, a.a_sst = a_stat.sb, a_stat.st
    a.a_gamma = _ptr(a_gamma)
    a.a_rowsum, a.a_rowsum2, a.a_dalpha = _ptr(a_rowsum), _ptr(a_rowsum2), _ptr(a_dalpha)
    a.B, a.ldb, a.bbs = _ptr(B), B.stride(1), B.stride(0)
    a.bflags = bflags
    # the library's wgrad kernel works on 64 x 64 output tiles, 64 x 128 for the NORMBWD variant when N allows,
    # 128 x 64 for the other variants when M allows
    tn = 128 if (aflags & XF_NORMBWD) and N % 128 == 0 else 64
    tm = 128 if not (aflags & XF_NORMBWD) and M % 128 == 0 else 64
    tiles = ((M + tm - 1) // tm) * ((N + tn - 1) // tn)
    # one generation of resident blocks: 4 per CU (3 for the 64 x 128 and 128 x 64 tiles, whose LDS buffers are larger)
    a.kchunk = kchunk or pick_kchunk(T, batch, tiles, 768 if (tn == 128 or tm == 128) else 1024)
    a.b_alpha = _ptr(b_alpha)
    if b_stat is not None:
        a.b_mean, a.b_rstd, a.b_ssb, a.b_sst = _ptr(b_stat.mean), _ptr(b_stat.rstd), b_stat.sb, b_stat.st
    a.b_gamma, a.b_beta = _ptr(b_gamma), _ptr(b_beta)
    a.a_rowsum_b, a.wdot_sums = _ptr(a_rowsum_b), _ptr(wdot_sums)
    if wdot is not None:
        a.wdot, a.ldwd = _ptr(wdot), wdot.stride(0)
    if wdot2 is not None:
        a.wdot2, a.ldwd2 = _ptr(wdot2), wdot2.stride(0)
    a.C, a.ldc = _ptr(Cout), ldc
    a.C2, a.ldc2 = _ptr(C2), ldc2
    stream = _stream()
    need = lib.ctn_gemm_nt_ws_floats(C.byref(a))
    if need > 0:
        ws = _nt_workspace(need, Cout.device, stream)
        if ws is not None:                     # (tests replace _nt_workspace to reach the atomics route)
            a.ws, a.ws_floats = _ptr(ws), ws.numel()
    _lib.check(lib.ctn_gemm_nt(C.byref(a), stream), "ctn_gemm_nt")


_nt_ws = {}


def _nt_workspace(need, device, stream):
    """The partial-output workspace of ctn_gemm_nt's whole-output kernels: one buffer per (device, stream), since calls
    on different streams may run at the same time; grown on demand.  Under stream capture the buffer comes from the
    graph's pool like every activation of the captured pass, and stays referenced here for the graph's replays."""
    key = (device.index, stream)
    ws = _nt_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=device, dtype=torch.float32)
        _nt_ws[key] = ws
    return ws


def heads_bwd_usable(M1, M2, H, T, stat, y2=None):
    """True when ctn_heads_bwd takes these head shapes in the current arithmetic (three bf16 pieces, gLN); y2, when
    given, is checked against the kernel's 32-bit byte offsets (the widest tensor it reads)."""
    if y2 is not None and ((y2.shape[0] - 1) * y2.stride(0) + H * y2.stride(1)) * 4 >= 0x7ffffff0:
        return False
    return (_gemm_split == 3 and H % 128 == 0 and M1 % 32 == 0 and M2 % 32 == 0 and M2 > 0 and (M1 + M2) in (128, 256)
            and T >= 4 and stat is not None and not stat.cln)


def heads_bwd(dY1, dY2, y2, alpha, stat, gamma, beta, W_prep, dn2, eacc, dW1, dW2, rowsum1, rowsum2):
    """The backward pass of a layer's two pointwise heads in one launch (ctn_heads_bwd in include/ctn.h): dn2 and its
    norm sums (into eacc), the weight gradients dW1 (M1, H) / dW2 (M2, H) and the bias gradients, from dY1 (B, M1, T) or
    None, dY2 (B, M2, T), the pre-activation y2 (B, H, T) and the prepared (M1 + M2) x H input-gradient matrix."""
    lib = _lib.load()
    a = _lib.HeadsBwdArgs()
    B, H, T = y2.shape
    a.batch, a.T, a.H = B, T, H
    a.M1, a.M2 = (0 if dY1 is None else dY1.shape[1]), dY2.shape[1]
    if dY1 is not None:
        a.dY1, a.lddy1, a.dy1bs = _ptr(dY1), dY1.stride(1), dY1.stride(0)
        a.dW1, a.lddw1 = _ptr(dW1), dW1.stride(0)
    a.dY2, a.lddy2, a.dy2bs = _ptr(dY2), dY2.stride(1), dY2.stride(0)
    a.dW2, a.lddw2 = _ptr(dW2), dW2.stride(0)
    a.y2, a.ldy2, a.y2bs = _ptr(y2), y2.stride(1), y2.stride(0)
    a.alpha = _ptr(alpha)
    a.mean, a.rstd, a.ssb = _ptr(stat.mean), _ptr(stat.rstd), stat.sb
    a.gamma, a.beta = _ptr(gamma), _ptr(beta)
    a.W_prep = _ptr(W_prep)
    a.dn2, a.lddn2, a.dn2bs = _ptr(dn2), dn2.stride(1), dn2.stride(0)
    a.esums = _ptr(eacc.sums)
    a.rowsum1, a.rowsum2 = _ptr(rowsum1), _ptr(rowsum2)
    _lib.check(lib.ctn_heads_bwd(C.byref(a), _stream()), "ctn_heads_bwd")


# ------------------------------------------------------------------------------------------------ depthwise
def dw_fwd(x, w, bias, y, dilation, pad_left, *, xflags=0, alpha_in=None, stat=None, gamma=None, beta=None,
           alpha_out=None, acc=None, stats=True, fin=None):
    lib = _lib.load()
    B, Cc, T = x.shape
    a = _lib.DwFwdArgs()
    a.batch, a.C, a.T, a.P = B, Cc, T, w.shape[-1]
    a.dilation, a.pad_left, a.xflags = dilation, pad_left, xflags
    a.cln = 1 if (acc is not None and acc.col is not None) else 0
    a.x, a.ldx, a.xbs = _ptr(x), x.stride(1), x.stride(0)
    a.alpha_in = _ptr(alpha_in)
    if stat is not None:
        a.mean, a.rstd, a.ssb, a.sst = _ptr(stat.mean), _ptr(stat.rstd), stat.sb, stat.st
    a.gamma, a.beta = _ptr(gamma), _ptr(beta)
    a.w, a.bias = _ptr(w), _ptr(bias)
    a.y, a.ldy, a.ybs = _ptr(y), y.stride(1), y.stride(0)
    if acc is not None:
        a.alpha_out = _ptr(alpha_out)
        a.esums = _ptr(acc.sums)
        a.ecol, a.ecol_ld = _ptr(acc.col), acc.ld
    _set_fin(a.fin, fin)
    _lib.check(lib.ctn_dw_fwd(C.byref(a), _stream()), "ctn_dw_fwd")


def dw_bwd(dn2, x, w, dw, dbias, dx, dilation, pad_left, *, xflags=0, alpha1=None, stat1=None, gamma1=None,
           beta1=None, dgamma1=None, dbeta1=None, iacc=None, y2=None, alpha2=None, stat2=None, gamma2=None,
           dgamma2=None, dbeta2=None, dalpha2=None, bias=None, recompute_y2=False, fin=None):
    lib = _lib.load()
    B, Cc, T = x.shape
    a = _lib.DwBwdArgs()
    a.batch, a.C, a.T, a.P = B, Cc, T, w.shape[-1]
    a.dilation, a.pad_left, a.xflags = dilation, pad_left, xflags
    a.oflags = XF_NORMBWD if stat2 is not None else 0
    cln = (stat1 is not None and stat1.cln) or (stat2 is not None and stat2.cln)
    a.cln = 1 if cln else 0
    a.dn2, a.lddn2, a.dn2bs = _ptr(dn2), dn2.stride(1), dn2.stride(0)
    if y2 is not None:
        a.y2, a.ldy2, a.y2bs = _ptr(y2), y2.stride(1), y2.stride(0)
    a.alpha2 = _ptr(alpha2)
    if stat2 is not None:
        a.mean2, a.rstd2, a.c1_2, a.c2_2 = _ptr(stat2.mean), _ptr(stat2.rstd), _ptr(stat2.c1), _ptr(stat2.c2)
        a.ssb2, a.sst2 = stat2.sb, stat2.st
    a.gamma2 = _ptr(gamma2)
    a.dgamma2, a.dbeta2, a.dalpha2 = _ptr(dgamma2), _ptr(dbeta2), _ptr(dalpha2)
    a.x, a.ldx, a.xbs = _ptr(x), x.stride(1), x.stride(0)
    a.alpha1 = _ptr(alpha1)
    if stat1 is not None:
        a.mean1, a.rstd1, a.ssb1, a.sst1 = _ptr(stat1.mean), _ptr(stat1.rstd), stat1.sb, stat1.st
    a.gamma1, a.beta1 = _ptr(gamma1), _ptr(beta1)
    a.w, a.dw, a.dbias = _ptr(w), _ptr(dw), _ptr(dbias)
    a.dx, a.lddx, a.dxbs = _ptr(dx), dx.stride(1), dx.stride(0)
    if iacc is not None:
        a.isums, a.icol, a.icol_ld = _ptr(iacc.sums), _ptr(iacc.col), iacc.ld
    a.dgamma1, a.dbeta1 = _ptr(dgamma1), _ptr(dbeta1)
    # whole-row gLN launches can recompute the convolution output from x instead of reading it (see include/ctn.h);
    # the library needs y2 for every other launch, so it is passed whenever there is one
    rows_fit = T <= 4096 and w.shape[-1] <= 4 and not cln and stat1 is not None and stat2 is not None
    a.bias, a.recompute_y2 = _ptr(bias), 1 if (recompute_y2 and rows_fit) else 0
    _set_fin(a.fin, fin)
    _lib.check(lib.ctn_dw_bwd(C.byref(a), _stream()), "ctn_dw_bwd")


# ------------------------------------------------------------------------------------------------ codec
def encoder_fwd(x2d, pad_left, W, S, relu, w_out, acc=None):
    """x2d (B, Tin) unit stride; W (N, 1, L); w_out act (B, N, Tf)."""
    lib = _lib.load()
    B, Tin = x2d.shape
    N, L = W.shape[0], W.shape[-1]
    Tf = w_out.shape[2]
    sums = col = None
    ld = 0
    if acc is not None:
        sums, col, ld = acc.sums, acc.col, acc.ld
    _lib.check(lib.ctn_encoder_fwd(_ptr(x2d), x2d.stride(0), Tin, pad_left, B, _ptr(W), N, L, S, 1 if relu else 0,
                                   _ptr(w_out), Tf, w_out.stride(1), _ptr(sums), _ptr(col), ld, _stream()),
               "ctn_encoder_fwd")


def encoder_bwd(x2d, pad_left, N, L, S, relu, w, dW, *, dw_a=None, dn=None, stat=None, gamma=None, dgamma=None,
                dbeta=None, dwt_out=None):
    lib = _lib.load()
    B, Tin = x2d.shape
    Tf = w.shape[2]
    st = stat
    _lib.check(lib.ctn_encoder_bwd(
        _ptr(x2d), x2d.stride(0), Tin, pad_left, B, N, L, S, 1 if relu else 0, _ptr(w), Tf, w.stride(1),
        _ptr(dw_a), dw_a.stride(1) if dw_a is not None else 0, _ptr(dn), dn.stride(1) if dn is not None else 0,
        _ptr(st.mean) if st else None, _ptr(st.rstd) if st else None, _ptr(st.c1) if st else None,
        _ptr(st.c2) if st else None, st.sb if st else 0, st.st if st else 0, _ptr(gamma), _ptr(dW), _ptr(dgamma),
        _ptr(dbeta), _ptr(dwt_out), dwt_out.stride(1) if dwt_out is not None else 0, _stream()), "ctn_encoder_bwd")


def maskdecode_fwd(w, mask, n_src, Wd, S, out, trim_left):
    """w act (B, N, Tf); mask act (B, n_src*N, Tf) or None; out (B, n_src, Tout) unit stride."""
    lib = _lib.load()
    B, N, Tf = w.shape
    L = Wd.shape[-1]
    _lib.check(lib.ctn_maskdecode_fwd(_ptr(w), w.stride(1), _ptr(mask), mask.stride(1) if mask is not None else 0, B,
                                      n_src, N, Tf, _ptr(Wd), L, S, _ptr(out), out.stride(1), out.stride(0),
                                      out.shape[2], trim_left, _stream()), "ctn_maskdecode_fwd")


def maskdecode_bwd(dout, trim_left, w, mask, n_src, Wd, S, sigmoid_bwd, dz, dw_dec, dWd):
    lib = _lib.load()
    B, N, Tf = w.shape
    L = Wd.shape[-1]
    _lib.check(lib.ctn_maskdecode_bwd(_ptr(dout), dout.stride(1), dout.stride(0), dout.shape[2], trim_left, _ptr(w),
                                      w.stride(1), _ptr(mask), mask.stride(1) if mask is not None else 0, B, n_src, N,
                                      Tf, _ptr(Wd), L, S, 1 if sigmoid_bwd else 0, _ptr(dz),
                                      dz.stride(1) if dz is not None else 0, _ptr(dw_dec),
                                      dw_dec.stride(1) if dw_dec is not None else 0, _ptr(dWd), _stream()),
               "ctn_maskdecode_bwd")


# ------------------------------------------------------------------------------------------------ loss / optimizer
def _check_signal_pair(est, tgt):
    """The loss kernels index the target with the estimate's sizes and strides: refuse anything else here, before a
    mismatched pair becomes a silent wrong loss or an out-of-bounds device read."""
    _require_cuda(est, tgt)
    # strides of size-1 dimensions are never multiplied by a non-zero index, so they may differ
    strides_differ = est.dim() == 3 and tgt.dim() == 3 and any(
        n > 1 and a != b for n, a, b in zip(est.shape, est.stride(), tgt.stride()))
    if (est.dim() != 3 or tgt.shape != est.shape or strides_differ or (est.shape[2] > 1 and est.stride(2) != 1)
            or est.dtype != torch.float32 or tgt.dtype != torch.float32 or tgt.device != est.device):
        raise _lib.CtnError("loss kernels need estimate and target as fp32 (B, S, T) tensors of the same shape, strides "
                            "and device with unit time stride; got {} {} and {} {}".format(
                                tuple(est.shape), est.stride(), tuple(tgt.shape), tgt.stride()))


def pit_sisdr_fwd(est, tgt, eps, fixed_perm=False):
    """est, tgt (B, S, T) unit time stride, same strides.  Returns pair, loss_b, perm (int64), coef."""
    lib = _lib.load()
    _check_signal_pair(est, tgt)
    B, S, T = est.shape
    dev = est.device
    work = torch.zeros((B, 2 * S * S + S), device=dev, dtype=torch.float64)
    pair = torch.empty((B, S, S), device=dev, dtype=torch.float32)
    loss_b = torch.empty((B,), device=dev, dtype=torch.float32)
    perm = torch.empty((B, S), device=dev, dtype=torch.int64)
    coef = torch.empty((B, S, 2), device=dev, dtype=torch.float32)
    _lib.check(lib.ctn_pit_sisdr_fwd(_ptr(est), _ptr(tgt), B, S, T, est.stride(1), est.stride(0), eps,
                                     1 if fixed_perm else 0, _ptr(work),
                                     _ptr(pair), _ptr(loss_b), _ptr(perm), _ptr(coef), _stream()),
               "ctn_pit_sisdr_fwd")
    return pair, loss_b, perm, coef


def pit_sisdr_bwd(est, tgt, perm, coef, gscale, gs_bstride, gs_istride):
    lib = _lib.load()
    _check_signal_pair(est, tgt)
    B, S, T = est.shape
    dest = torch.empty((B, S, T), device=est.device, dtype=torch.float32)
    _lib.check(lib.ctn_pit_sisdr_bwd(_ptr(est), _ptr(tgt), B, S, T, est.stride(1), est.stride(0), _ptr(perm),
                                     _ptr(coef), _ptr(gscale), gs_bstride, gs_istride, _ptr(dest), dest.stride(1), dest.stride(0),
                                     _stream()), "ctn_pit_sisdr_bwd")
    return dest


def dwconv_generic_fwd(x, w, bias, stride, dilation, pad_left, Tout):
    """x (B, C, Tin) with unit time stride, w (C, K) -> y (B, C, Tout): depthwise convolution of any geometry."""
    lib = _lib.load()
    B, Cc, Tin = x.shape
    y = new_act(B, Cc, Tout, x.device)
    _lib.check(lib.ctn_dwconv_generic_fwd(_ptr(x), x.stride(0), x.stride(1), B, Cc, Tin, _ptr(w), _ptr(bias), w.shape[1],
                                          stride, dilation, pad_left, _ptr(y), y.stride(0), y.stride(1), Tout, _stream()),
               "ctn_dwconv_generic_fwd")
    return y


def dwconv_generic_bwd(x, w, dy, stride, dilation, pad_left, need_dx, need_db):
    lib = _lib.load()
    B, Cc, Tin = x.shape
    Tout = dy.shape[2]
    dx = new_act(B, Cc, Tin, x.device) if need_dx else None
    dW = torch.zeros_like(w)
    db = torch.zeros((Cc,), device=x.device, dtype=torch.float32) if need_db else None
    _lib.check(lib.ctn_dwconv_generic_bwd(_ptr(x), x.stride(0), x.stride(1), B, Cc, Tin, _ptr(w), w.shape[1], stride,
                                          dilation, pad_left, _ptr(dy), dy.stride(0), dy.stride(1), Tout, _ptr(dx),
                                          dx.stride(0) if need_dx else 0, dx.stride(1) if need_dx else 0, _ptr(dW),
                                          _ptr(db), _stream()), "ctn_dwconv_generic_bwd")
    return dx, dW, db


PAIR_SISDR, PAIR_SDR = 0, 1


def pair_sdr_fwd(est, tgt, eps, kind):
    """est, tgt (B, S, T) contiguous, S <= 4.  Returns pair (B, S, S) = criterion(est_i, tgt_j) in dB (kind: PAIR_SISDR or
    PAIR_SDR) and paircoef (B, S, S, 2), the coefficients of d pair / d est_i = ct * tgt_j + ce * est_i."""
    lib = _lib.load()
    _check_signal_pair(est, tgt)
    B, S, T = est.shape
    dev = est.device
    work = torch.zeros((B, 2 * S * S + S), device=dev, dtype=torch.float64)
    pair = torch.empty((B, S, S), device=dev, dtype=torch.float32)
    paircoef = torch.empty((B, S, S, 2), device=dev, dtype=torch.float32)
    _lib.check(lib.ctn_pair_sdr_fwd(_ptr(est), _ptr(tgt), B, S, T, est.stride(1), est.stride(0), eps, int(kind),
                                    _ptr(work), _ptr(pair), _ptr(paircoef), _stream()), "ctn_pair_sdr_fwd")
    return pair, paircoef


def pair_sdr_bwd(est, tgt, A, Cc):
    """dest_i = sum_j A[b, i, j] tgt_j + Cc[b, i] est_i."""
    lib = _lib.load()
    _check_signal_pair(est, tgt)
    B, S, T = est.shape
    dest = torch.empty((B, S, T), device=est.device, dtype=torch.float32)
    _lib.check(lib.ctn_pair_sdr_bwd(_ptr(est), _ptr(tgt), B, S, T, est.stride(1), est.stride(0), _ptr(A), _ptr(Cc),
                                    _ptr(dest), dest.stride(1), dest.stride(0), _stream()), "ctn_pair_sdr_bwd")
    return dest


def sumsq(g, acc):
    lib = _lib.load()
    _lib.check(lib.ctn_sumsq(_ptr(g), g.numel(), _ptr(acc), _stream()), "ctn_sumsq")


def adam_step(p, g, m, v, sumsq_acc, max_norm, grad_scale, lr, beta1, beta2, eps, step):
    lib = _lib.load()
    _lib.check(lib.ctn_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(sumsq_acc),
                                 float(max_norm or 0.0), float(grad_scale), lr, beta1, beta2, eps, step, _stream()),
               "ctn_adam_step")
