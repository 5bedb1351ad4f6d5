"""Forward/backward schedules of the Conv-TasNet hot path over the HIP kernels, and the
torch.autograd.Function wrappers the nn.Modules call.

Each schedule is a fixed sequence of C-ABI launches (see include/ctn.h); PyTorch only supplies
device memory, the current stream and the autograd graph edges.  Data flow of one residual layer
(reference: src/models/tdcn.py:107-147 with :177-196 inlined), B = bottleneck, H = hidden channels:

  forward   x (B) --gemm_nn[STATS]--> y1 (H, pre-PReLU) --dw_fwd[PReLU+norm on load, stats out]--> y2 (H)
            --gemm_nn[PReLU+norm on load, HEADS]--> out = . + x (B),  skip_sum += . (Sc)
  backward  (dout, dskip) --gemm_nn[NORMBWD epilogue]--> dn2 ; gemm_nt -> dWo, dWs, dbo, dbs
            dn2 --dw_bwd--> dn1 (+ dwd, dbd, dgamma/dbeta 1 and 2, dalpha2)
            dn1 --gemm_nn[NORMBWD on load, ADD dout]--> dx ; gemm_nt[NORMBWD on load] -> dW1, db1, dalpha1

Only y1, y2 and the per-norm statistics are kept for backward; normalised tensors never exist in HBM.
"""
import collections
import os

import torch

from . import ops
from .ops import (XF_PRELU, XF_NORM, XF_NORMBWD, EPI_STORE, EPI_STATS, EPI_HEADS, EPI_SIGMOID, EPI_PRELUBWD,
                  EPI_NORMBWD, EPI_ADD, Accum, new_act, as_act)


# ------------------------------------------------------------------------------------------------ helpers
# The depthwise backward kernel recomputes the convolution output from the layer's first pre-activation instead of
# reading the stored one (three tensors over HBM instead of four).  False restores the reading form (diagnostics, tests).
RECOMPUTE_DW_OUTPUT = os.environ.get("CTN_DW_RECOMPUTE", "1") != "0"

# In the default arithmetic (three bf16 pieces) the weight operand of every pointwise GEMM can be split once per step
# into the matrix instruction's own layout (ops.prepared / PreparedWeights) instead of by every wave that reads it; the
# results are bit-identical.  On by default: with the weights arriving split, the waves of a GEMM block sit side by
# side along time and each splits only its own activation fragment (a quarter of the vector work of the in-kernel
# split), which is 4-5 % of a training step at the benchmark's batch.  CTN_PREPARED_WEIGHTS=0 restores the in-kernel
# split of both operands.
USE_PREPARED_WEIGHTS = os.environ.get("CTN_PREPARED_WEIGHTS", "1") != "0"

# The backward coefficients of the norm in front of a residual layer's heads can be derived from by-products of the heads'
# weight-gradient launch instead of a statistics epilogue (and a read of the pre-activation tensor) in the heads'
# input-gradient GEMM (gLN only; ctn_heads_bwd_coefs): 3 % of a training step at the benchmark's batch.  Opt-in
# (CTN_HEADS_COEFS_FROM_WGRAD=1): every tensor-valued gradient agrees with float64 as before (1.5e-6 of its largest
# element at the paper width), but the scalar PReLU-slope gradients -- sums in which the three parts of the norm's
# backward formula cancel -- lose a digit in the three-piece arithmetic (worst of 49 at the paper width, 4 s inputs:
# 2.6e-3 relative against 1.5e-4 with the epilogue's sums, which are taken from the very tensor the formula is then
# applied to; with f32 MFMAs both routes are at 1e-4).  tools/diag_heads_coefs.py measures it.
HEADS_COEFS_FROM_WGRAD = os.environ.get("CTN_HEADS_COEFS_FROM_WGRAD", "0") != "0"

# The backward pass of a layer's two heads as ONE launch (ctn_heads_bwd: the input gradient with its norm sums and the
# weight / bias gradients share one pass over dY and y2) instead of a ctn_gemm_nn and a ctn_gemm_nt launch.  gLN, the
# three-piece arithmetic and prepared weights only.  Opt-in (CTN_FUSED_HEADS_BWD=1): it moves a third less data, but
# with one 142 KB block per CU its phases do not overlap and the launch takes 598 us against 463 us for the pair inside
# a step at the benchmark's batch (35.3 against 34.7 ms per step; analysis in csrc/ctn_heads_bwd.hip).
FUSED_HEADS_BWD = os.environ.get("CTN_FUSED_HEADS_BWD", "0") != "0"

_ZERO_ROWS = {}


def _zero_row(n, dev):
    """A shared all-zero float row of n elements (a beta that is not there)."""
    key = (n, str(dev))
    if key not in _ZERO_ROWS:
        _ZERO_ROWS[key] = torch.zeros(n, device=dev, dtype=torch.float32)
    return _ZERO_ROWS[key]


FLAT_ALIGN = 64      # floats: every tensor of a flat parameter / gradient buffer starts on a 256-byte boundary


def flat_layout(params):
    """Offsets (in floats) of `params` inside a flat buffer and the buffer's length.  Scalars (PReLU slopes)
    would otherwise leave every following weight matrix 4-byte aligned only, which takes the GEMMs that read
    weights or write weight gradients in place off their 16-byte paths."""
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN
    return offsets, off


class GradSink:
    """Zeroed gradient buffers for a list of parameters (one flat allocation, views per parameter).

    PReLU slopes are reduced in fp64 by the kernels; their slots live in `dalpha` and are folded
    into the fp32 views by finish()."""

    def __init__(self, params, flat=None, zero=True):
        self.params = list(params)
        dev = self.params[0].device
        self.offsets, total = flat_layout(self.params)
        if flat is None:
            self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        else:                                  # caller-owned buffer (e.g. the trainer's all-reduce bucket)
            assert flat.numel() == total and flat.dtype == torch.float32 and flat.is_contiguous()
            self.flat = flat
            if zero:                           # zero=False: the caller zeroed it and several sinks add into it
                self.flat.zero_()
        self.views = [self.flat[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets)]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._alpha = []
        self._folded = set()
        self.dalpha = None

    def g(self, p):
        return None if p is None else self.views[self._index[id(p)]]

    def alpha_slot(self, p):
        """fp64 slot-replicated accumulator (STAT_SLOTS doubles) for PReLU slope p."""
        if self.dalpha is None:
            self.dalpha = torch.zeros((len(self.params), ops._lib.STAT_SLOTS), device=self.flat.device,
                                      dtype=torch.float64)
        i = self._index[id(p)]
        self._alpha.append(i)
        return self.dalpha[i]

    _index_cache = {}      # (device, slope indices, offsets) -> device index tensors; built once, so that a step
                           # has no host-to-device copy (a synchronisation point, and not capturable in a graph)

    def fold_alphas(self, lo=0, hi=None):
        """Adds the fp64 slope-gradient slots of the parameters with index in [lo, hi) that were handed out and not
        folded yet into their fp32 views.  A region of the flat buffer is final -- ready to be all-reduced -- once every
        kernel that adds to it has been enqueued and its slopes are folded."""
        hi = len(self.params) if hi is None else hi
        idx = tuple(sorted(i for i in set(self._alpha) if lo <= i < hi and i not in self._folded))
        if not idx:
            return
        self._folded.update(idx)
        dev = self.flat.device
        key = (str(dev), idx, tuple(self.offsets[i] for i in idx))
        hit = GradSink._index_cache.get(key)
        if hit is None:
            hit = (torch.tensor(key[1], device=dev, dtype=torch.long),
                   torch.tensor(key[2], device=dev, dtype=torch.long))
            GradSink._index_cache[key] = hit
        src, dst = hit
        self.flat.index_add_(0, dst, self.dalpha.index_select(0, src).sum(dim=1).float())

    def finish(self):
        self.fold_alphas()
        return self.views


def _flat(p):
    return None if p is None else p.detach().reshape(-1)


# Prepared matrices made outside a PreparedWeights store (module-by-module passes: validation, ResidualBlock1d or
# Conv1d on their own) are kept per source tensor and reused while the sources have not changed: keyed on address,
# layout and torch's version counter, which every in-place update through torch bumps (optimizers, load_state_dict,
# copy_); at most _PREP_CACHE_MAX matrices, least recently used first out.  An update that writes through a raw pointer does not -- FlatTrainer's fused Adam kernel -- and must call
# invalidate_prepared_cache(), which FlatTrainer.optimizer_step does.
_PREP_CACHE = collections.OrderedDict()
_PREP_CACHE_MAX = 512


def invalidate_prepared_cache():
    _PREP_CACHE.clear()


def _prepared(store, key, K, M, parts, dev, K1=None):
    """The prepared form of a K x M weight operand (parts: [(2-D source view, transpose, k_off, m_off)]) or None when
    this arithmetic / shape does not take one.  `store` (a dict of matrices prepared earlier in the step) is used when
    it has the key; otherwise the matrix comes from the per-tensor cache above or is prepared now."""
    if not (USE_PREPARED_WEIGHTS and ops.prep_usable(M, K, K1)):
        return None
    if store is not None and store.get(key) is not None:
        return store[key]
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():      # a graph's private pool must not
                                                                                    # leak into the cache
        return ops.prepared(K, M, parts, dev)
    ck = (K, M, str(dev)) + tuple((src.data_ptr(), src._version, tuple(src.shape), src.stride(), bool(tr), ko, mo)
                                  for src, tr, ko, mo in parts)
    hit = _PREP_CACHE.get(ck)
    if hit is not None:
        _PREP_CACHE.move_to_end(ck)
        return hit[0]
    out = ops.prepared(K, M, parts, dev)
    # the entry holds the source views: their storage cannot be freed and its address handed to another tensor (with an
    # equal version count) while the key is in the cache
    _PREP_CACHE[ck] = (out, [src for src, _, _, _ in parts])
    while len(_PREP_CACHE) > _PREP_CACHE_MAX:
        _PREP_CACHE.popitem(last=False)
    return out


def _layer_parts(L):
    """key -> (K, M, K1, parts) of the four weight operands of one residual layer's GEMMs."""
    H, Bc, Sc = L.H, L.Bc, L.Sc
    Mo = Bc if L.dual else 0
    W1, Ws = L.W1.view(H, Bc), L.Ws.view(Sc, H)
    heads_f = ([(L.Wo.view(Bc, H), True, 0, 0)] if L.dual else []) + [(Ws, True, 0, Mo)]
    heads_d = ([(L.Wo.view(Bc, H), False, 0, 0)] if L.dual else []) + [(Ws, False, Mo, 0)]
    return {"W1f": (Bc, H, None, [(W1, True, 0, 0)]),                 # bottleneck forward: A[k = in][m = out]
            "Hf": (H, Mo + Sc, None, heads_f),                        # both heads forward, side by side along m
            "Hd": (Mo + Sc, H, Mo if L.dual else None, heads_d),      # both heads' input gradient, stacked along k
            "W1d": (H, Bc, None, [(W1, False, 0, 0)])}                # bottleneck input gradient


class PreparedWeights:
    """The weight operands of the separator's pointwise GEMMs, laid out once per optimizer step instead of once per use:
    k-major fp32 copies of every weight the forward GEMMs read (ctn_transpose_batch: two launches instead of 73) and,
    when `planes` is asked for, every weight operand in prepared form (ctn_weight_prep) in one bf16 buffer.  refresh()
    runs on the stream the pass starts from.  The parameters must keep their addresses (views of a flat buffer)."""

    def __init__(self, mp, planes=False):
        dev = mp.Wb.device
        self.layers = [dict() for _ in mp.layers]
        self.sep = {}
        # ---- k-major fp32: W1t (Bc, H) and the two heads side by side (H, Mo + Sc) per layer; Wbt, Wmt
        entries, self._keep = [], []

        def kmajor(store, key, K, M, pieces):          # pieces: [(source (m, K), column offset)]
            dst = torch.empty((K, M), device=dev, dtype=torch.float32)
            store[key] = dst
            for src, coff in pieces:
                src = src.contiguous()
                self._keep.append(src)
                entries.append((src, dst, M, coff))

        for i, (L, _) in enumerate(mp.layers):
            H, Bc, Sc = L.H, L.Bc, L.Sc
            Mo = Bc if L.dual else 0
            kmajor(self.layers[i], "W1t", Bc, H, [(L.W1.view(H, Bc), 0)])
            kmajor(self.layers[i], "Wct", H, Mo + Sc,
                   ([(L.Wo.view(Bc, H), 0)] if L.dual else []) + [(L.Ws.view(Sc, H), Mo)])
        Bc, N = mp.Wb.shape[0], mp.Wb.shape[1]
        Mm, Sc = mp.Wm.shape[0], mp.Wm.shape[1]
        kmajor(self.sep, "Wbt", N, Bc, [(mp.Wb.detach().view(Bc, N), 0)])
        kmajor(self.sep, "Wmt", Sc, Mm, [(mp.Wm.detach().view(Mm, Sc), 0)])
        self.t_items = ops.transpose_items(entries)
        # ---- prepared planes
        self.items = None
        self.planes = bool(planes)             # whether they were asked for (no matrix may qualify: items stays None)
        if planes:
            specs = []
            for i, (L, _) in enumerate(mp.layers):
                for key, (K, M, K1, parts) in _layer_parts(L).items():
                    specs.append((self.layers[i], key, K, M, K1, parts))
            for key, (K, M, K1, parts) in _separator_parts(mp).items():
                specs.append((self.sep, key, K, M, K1, parts))
            specs = [sp for sp in specs if sp[3] > 64 and sp[3] % 4 == 0 and sp[2] % 16 == 0 and sp[2] <= 4096 and
                     (sp[4] or 0) % 16 == 0]
            total = sum(ops.weight_planes_numel(K, M) for _, _, K, M, _, _ in specs)
            self.buf = torch.zeros(max(total, 8), device=dev, dtype=torch.bfloat16)
            pentries, off = [], 0
            for store, key, K, M, K1, parts in specs:
                n = ops.weight_planes_numel(K, M)
                dst = self.buf[off:off + n]
                off += n
                store[key] = dst
                pentries += [(src, tr, ko, mo, K, M, dst) for src, tr, ko, mo in parts]
            self._keep += [e[0] for e in pentries]
            self.items = ops.wprep_items(pentries) if pentries else None

    def refresh(self):
        ops.transpose_batch(self.t_items)
        if self.items is not None:
            ops.weight_prep(self.items)


class Layer:
    """Detached, flattened parameter views of one ResidualBlock1d (+ its tdcn DepthwiseSeparableConv1d)."""

    NAMES_DUAL = ["W1", "b1", "a1", "g1", "be1", "wd", "bd", "a2", "g2", "be2", "Wo", "bo", "Ws", "bs"]
    NAMES_SINGLE = ["W1", "b1", "a1", "g1", "be1", "wd", "bd", "a2", "g2", "be2", "Ws", "bs"]

    def __init__(self, tensors, dual):
        names = self.NAMES_DUAL if dual else self.NAMES_SINGLE
        assert len(tensors) == len(names)
        self.dual = dual
        self.orig = dict(zip(names, tensors))
        for n in self.NAMES_DUAL:
            setattr(self, n, None)
        for n, t in zip(names, tensors):
            setattr(self, n, t.detach())
        self.H, self.Bc = self.W1.shape[0], self.W1.shape[1]
        self.Sc = self.Ws.shape[0]
        self.P = self.wd.shape[-1]
        self.planes = None           # dict of prepared weight operands (PreparedWeights.layers[i]) or None


def _pads(P, dilation, causal):
    padding = (P - 1) * dilation                      # src/models/tdcn.py:123 with stride 1
    return padding if causal else padding // 2        # :125-130


# ------------------------------------------------------------------------------------------------ residual layer
def layer_forward(L, x, skip, accumulate, dilation, causal, eps):
    """x act (B, Bc, T); skip act (B, Sc, T) written (accumulate=False) or added to.  Returns out, saved."""
    B, _, T = x.shape
    dev = x.device
    H, Bc, Sc, P = L.H, L.Bc, L.Sc, L.P
    # bottleneck 1x1 + PReLU statistics
    parts = _layer_parts(L)
    W1p = _prepared(L.planes, "W1f", *parts["W1f"][:2], parts["W1f"][3], dev)
    W1t = None
    if W1p is None:
        W1t = L.planes.get("W1t") if L.planes is not None else None
        if W1t is None:
            W1t = torch.empty((Bc, H), device=dev, dtype=torch.float32)
            ops.transpose(L.W1.view(H, Bc), W1t, H)
    y1 = new_act(B, H, T, dev)
    acc1 = Accum(B, T, causal, dev)
    st1, fin1 = ops.stats_request(acc1, B, H, T, eps, dev)
    ops.gemm_nn(H, T, Bc, B, W1t, H, x, y1, bias=L.b1, emode=EPI_STATS, ealpha=L.a1, eacc=acc1, A_prep=W1p, fin=fin1)
    if st1 is None:
        st1 = ops.finalize_stats(acc1, B, H, T, eps, dev)
    # depthwise
    y2 = new_act(B, H, T, dev)
    acc2 = Accum(B, T, causal, dev)
    st2, fin2 = ops.stats_request(acc2, B, H, T, eps, dev, producer="dw")
    ops.dw_fwd(y1, L.wd.view(H, P), L.bd, y2, dilation, _pads(P, dilation, causal), xflags=XF_PRELU | XF_NORM,
               alpha_in=L.a1, stat=st1, gamma=L.g1.view(-1), beta=L.be1.view(-1), alpha_out=L.a2, acc=acc2, fin=fin2)
    if st2 is None:
        st2 = ops.finalize_stats(acc2, B, H, T, eps, dev)
    # heads
    Mo = Bc if L.dual else 0
    Mt = Mo + Sc
    Wcp = _prepared(L.planes, "Hf", *parts["Hf"][:2], parts["Hf"][3], dev)
    Wct = None
    if Wcp is None:
        Wct = L.planes.get("Wct") if L.planes is not None else None
        if Wct is None:
            Wct = torch.empty((H, Mt), device=dev, dtype=torch.float32)
            if L.dual:
                ops.transpose(L.Wo.view(Bc, H), Wct, Mt, 0)
            ops.transpose(L.Ws.view(Sc, H), Wct, Mt, Mo)
    out = new_act(B, Bc, T, dev) if L.dual else None
    ops.gemm_nn(Mt, T, H, B, Wct, Mt, y2, out if L.dual else skip, xflags=XF_PRELU | XF_NORM, alpha=L.a2, stat=st2,
                gamma=L.g2.view(-1), beta=L.be2.view(-1), bias=L.bo, bias2=L.bs, emode=EPI_HEADS, Mo=Mo,
                R=x if L.dual else None, Y2=skip, accumulate=accumulate, A_prep=Wcp)
    return out, (x, y1, y2, st1, st2)


def layer_backward(L, saved, dout, dskip, sink, dilation, causal, eps):
    """dout act (B, Bc, T) or None; dskip act (B, Sc, T).  Accumulates parameter grads, returns dx."""
    x, y1, y2, st1, st2 = saved
    B, _, T = x.shape
    dev = x.device
    H, Bc, Sc, P = L.H, L.Bc, L.Sc, L.P
    o = L.orig
    dual = L.dual and dout is not None
    dn2 = new_act(B, H, T, dev)
    parts = _layer_parts(L)
    M1 = Bc if dual else 0
    if HEADS_COEFS_FROM_WGRAD and not causal and ops.heads_coefs_usable(M1 + Sc, M1, H):
        # The norm-2 backward coefficients come out of the heads' WEIGHT gradient: sum gamma dn xhat = sum W * (the per-
        # sample weight gradient taken without beta), sum gamma dn = sum_c gamma_c sum_m W[m][c] rowsum[m] (derivation at
        # ctn_heads_bwd_coefs in include/ctn.h).  The input-gradient GEMM then stores its tile and nothing else: no
        # read of y2, no statistics in its epilogue.
        rs_b = ops.zeros((B, M1 + Sc), dev, torch.float32)
        wsum = ops.zeros((B, ops._lib.STAT_SLOTS), dev, torch.float64)
        Wo2 = L.Wo.view(Bc, H) if dual else None
        Ws2 = L.Ws.view(Sc, H)
        dWo = sink.g(o["Wo"]).view(Bc, H) if dual else None
        dWs = sink.g(o["Ws"]).view(Sc, H)
        zb = _zero_row(H, dev)
        if dual:
            ops.gemm_nt(Bc + Sc, H, T, B, dout, y2, dWo, H, Ab=dskip, M1=Bc, C2=dWs, ldc2=H, a_rowsum=sink.g(o["bo"]),
                        a_rowsum2=sink.g(o["bs"]), bflags=XF_PRELU | XF_NORM, b_alpha=L.a2, b_stat=st2,
                        b_gamma=L.g2.view(-1), b_beta=zb, a_rowsum_b=rs_b, wdot=Wo2, wdot2=Ws2, wdot_sums=wsum)
        else:
            ops.gemm_nt(Sc, H, T, B, dskip, y2, dWs, H, a_rowsum=sink.g(o["bs"]), bflags=XF_PRELU | XF_NORM,
                        b_alpha=L.a2, b_stat=st2, b_gamma=L.g2.view(-1), b_beta=zb, a_rowsum_b=rs_b, wdot=Ws2,
                        wdot_sums=wsum)
        ops.heads_bwd_coefs(Wo2 if dual else Ws2, Ws2 if dual else None, L.g2.view(-1), L.be2.view(-1), rs_b, wsum, st2,
                            B, H, T, dWo if dual else dWs, dWs if dual else None)
        if dual:
            Hdp = _prepared(L.planes, "Hd", *parts["Hd"][:2], parts["Hd"][3], dev, K1=Bc)
            ops.gemm_nn(H, T, Bc + Sc, B, Wo2, H, dout, dn2, A2=Ws2, lda2=H, K1=Bc, Xb=dskip, emode=EPI_STORE, A_prep=Hdp)
        else:
            Hdp = None if L.dual else _prepared(L.planes, "Hd", *parts["Hd"][:2], parts["Hd"][3], dev)
            ops.gemm_nn(H, T, Sc, B, Ws2, H, dskip, dn2, emode=EPI_STORE, A_prep=Hdp)
        return _layer_backward_tail(L, saved, dn2, dout, dual, sink, dilation, causal, eps, parts)
    if FUSED_HEADS_BWD and not causal and dual == L.dual and ops.heads_bwd_usable(M1, Sc, H, T, st2, y2) \
            and "gemm" not in ops.FOLD_FINISH:
        # one launch for the heads' whole backward pass: input gradient with its norm sums, weight and bias gradients
        Hdp = _prepared(L.planes, "Hd", *parts["Hd"][:2], parts["Hd"][3], dev, K1=Bc if dual else None)
        if Hdp is not None:
            bacc2 = Accum(B, T, causal, dev)
            ops.heads_bwd(dout if dual else None, dskip, y2, L.a2, st2, L.g2.view(-1), L.be2.view(-1), Hdp, dn2, bacc2,
                          sink.g(o["Wo"]).view(Bc, H) if dual else None, sink.g(o["Ws"]).view(Sc, H),
                          sink.g(o["bo"]) if dual else None, sink.g(o["bs"]))
            ops.bwd_coefs(bacc2, st2, B, H, T, eps)
            return _layer_backward_tail(L, saved, dn2, dout, dual, sink, dilation, causal, eps, parts)
    # heads: input gradient with the norm-2 backward sums in the epilogue
    bacc2 = Accum(B, T, causal, dev)
    finb2 = ops.coefs_request(bacc2, st2, B, H, T)
    if dual:
        Hdp = _prepared(L.planes, "Hd", *parts["Hd"][:2], parts["Hd"][3], dev, K1=Bc)
        ops.gemm_nn(H, T, Bc + Sc, B, L.Wo.view(Bc, H), H, dout, dn2, A2=L.Ws.view(Sc, H), lda2=H, K1=Bc, Xb=dskip,
                    emode=EPI_NORMBWD, E=y2, ealpha=L.a2, estat=st2, egamma=L.g2.view(-1), eacc=bacc2, A_prep=Hdp,
                    fin=finb2)
        ops.gemm_nt(Bc + Sc, H, T, B, dout, y2, sink.g(o["Wo"]).view(Bc, H), H, Ab=dskip, M1=Bc,
                    C2=sink.g(o["Ws"]).view(Sc, H), ldc2=H, a_rowsum=sink.g(o["bo"]), a_rowsum2=sink.g(o["bs"]),
                    bflags=XF_PRELU | XF_NORM, b_alpha=L.a2, b_stat=st2, b_gamma=L.g2.view(-1), b_beta=L.be2.view(-1))
    else:
        # (a dual layer that received no residual gradient: its prepared matrix stacks both heads and does not apply)
        Hdp = None if L.dual else _prepared(L.planes, "Hd", *parts["Hd"][:2], parts["Hd"][3], dev)
        ops.gemm_nn(H, T, Sc, B, L.Ws.view(Sc, H), H, dskip, dn2, emode=EPI_NORMBWD, E=y2, ealpha=L.a2, estat=st2,
                    egamma=L.g2.view(-1), eacc=bacc2, A_prep=Hdp, fin=finb2)
        ops.gemm_nt(Sc, H, T, B, dskip, y2, sink.g(o["Ws"]).view(Sc, H), H, a_rowsum=sink.g(o["bs"]),
                    bflags=XF_PRELU | XF_NORM, b_alpha=L.a2, b_stat=st2, b_gamma=L.g2.view(-1), b_beta=L.be2.view(-1))
    if finb2 is None:
        ops.bwd_coefs(bacc2, st2, B, H, T, eps)
    return _layer_backward_tail(L, saved, dn2, dout, dual, sink, dilation, causal, eps, parts)


def _layer_backward_tail(L, saved, dn2, dout, dual, sink, dilation, causal, eps, parts):
    """layer_backward from the depthwise convolution down: dn2 (gradient at the norm-2 output) and st2.c1 / c2 are in."""
    x, y1, y2, st1, st2 = saved
    B, _, T = x.shape
    dev = x.device
    H, Bc, Sc, P = L.H, L.Bc, L.Sc, L.P
    o = L.orig
    # depthwise backward (+ norm-2 / PReLU-2 backward on load, norm-1 sums on the way out)
    dn1 = new_act(B, H, T, dev)
    bacc1 = Accum(B, T, causal, dev)
    finb1 = ops.coefs_request(bacc1, st1, B, H, T, producer="dw")
    ops.dw_bwd(dn2, y1, L.wd.view(H, P), sink.g(o["wd"]).view(H, P), sink.g(o["bd"]), dn1, dilation,
               _pads(P, dilation, causal), xflags=XF_PRELU | XF_NORM, alpha1=L.a1, stat1=st1, gamma1=L.g1.view(-1),
               beta1=L.be1.view(-1), dgamma1=sink.g(o["g1"]).view(-1), dbeta1=sink.g(o["be1"]).view(-1), iacc=bacc1,
               y2=y2, alpha2=L.a2, stat2=st2, gamma2=L.g2.view(-1), dgamma2=sink.g(o["g2"]).view(-1),
               dbeta2=sink.g(o["be2"]).view(-1), dalpha2=sink.alpha_slot(o["a2"]), bias=L.bd, recompute_y2=RECOMPUTE_DW_OUTPUT,
               fin=finb1)
    if finb1 is None:
        ops.bwd_coefs(bacc1, st1, B, H, T, eps)
    # bottleneck: input gradient (+ residual path) and weight gradient, norm-1 / PReLU-1 backward on load
    dx = new_act(B, Bc, T, dev)
    W1dp = _prepared(L.planes, "W1d", *parts["W1d"][:2], parts["W1d"][3], dev)
    ops.gemm_nn(Bc, T, H, B, L.W1.view(H, Bc), Bc, dn1, dx, xflags=XF_NORMBWD, X2=y1, alpha=L.a1, stat=st1,
                gamma=L.g1.view(-1), emode=EPI_ADD, R=dout if dual else None, A_prep=W1dp)
    ops.gemm_nt(H, Bc, T, B, dn1, x, sink.g(o["W1"]).view(H, Bc), Bc, aflags=XF_NORMBWD, A2=y1, a_alpha=L.a1,
                a_stat=st1, a_gamma=L.g1.view(-1), a_rowsum=sink.g(o["b1"]), a_dalpha=sink.alpha_slot(o["a1"]))
    return dx


class ResidualBlockFn(torch.autograd.Function):
    """ResidualBlock1d.forward (src/models/tdcn.py:107-147): returns (output or None, skip)."""

    @staticmethod
    def forward(ctx, x, dilation, causal, dual, eps, *params):
        x = as_act(x.detach())
        L = Layer(params, dual)
        B, _, T = x.shape
        skip = new_act(B, L.Sc, T, x.device)
        out, saved = layer_forward(L, x, skip, False, dilation, causal, eps)
        ctx.L, ctx.saved_acts, ctx.cfg = L, saved, (dilation, causal, eps)
        ctx.params = params
        if dual:
            return out, skip
        return None, skip

    @staticmethod
    def backward(ctx, dout, dskip):
        dilation, causal, eps = ctx.cfg
        L = ctx.L
        x = ctx.saved_acts[0]
        B, _, T = x.shape
        if dskip is None:
            dskip = new_act(B, L.Sc, T, x.device, zero=True)
        dskip = as_act(dskip)
        dout = as_act(dout) if dout is not None else None
        sink = GradSink(ctx.params)
        dx = layer_backward(L, ctx.saved_acts, dout, dskip, sink, dilation, causal, eps)
        grads = sink.finish()
        return (dx, None, None, None, None) + tuple(grads)


# ------------------------------------------------------------------------------------------------ standalone norm
class LayerNormFn(torch.autograd.Function):
    """GlobalLayerNorm / CumulativeLayerNorm1d forward+backward (src/modules/norm.py:20-29, :58-95)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, causal):
        shape = x.shape
        x3 = as_act(x.detach().reshape(shape[0], shape[1], -1))
        B, Cc, T = x3.shape
        acc = Accum(B, T, causal, x3.device)
        ops.norm_stats(x3, None, acc)
        st = ops.finalize_stats(acc, B, Cc, T, eps, x3.device)
        y = new_act(B, Cc, T, x3.device)
        ops.norm_apply(x3, None, st, gamma.detach().reshape(-1), beta.detach().reshape(-1), y)
        ctx.saved = (x3, st, gamma, beta, eps, causal, shape)
        return y.reshape(shape) if len(shape) != 3 else y

    @staticmethod
    def backward(ctx, dy):
        x3, st, gamma, beta, eps, causal, shape = ctx.saved
        B, Cc, T = x3.shape
        dy3 = as_act(dy.reshape(B, Cc, T))
        g = gamma.detach().reshape(-1)
        dgamma = torch.zeros_like(g)
        dbeta = torch.zeros_like(g)
        acc = Accum(B, T, causal, x3.device)
        ops.norm_bwd_reduce(dy3, x3, None, st, g, acc, dgamma, dbeta)
        ops.bwd_coefs(acc, st, B, Cc, T, eps)
        dx = new_act(B, Cc, T, x3.device)
        ops.norm_bwd_apply(dy3, x3, None, st, g, dx, None)
        dx = dx.reshape(shape) if len(shape) != 3 else dx
        return dx, dgamma.view(gamma.shape), dbeta.view(beta.shape), None, None


# ------------------------------------------------------------------------------------------------ recurrence
class LinearFn(torch.autograd.Function):
    """nn.Linear over (rows, features) (src/models/dprnn.py:62,87 and :116,139) on the row-product kernels: forward and
    input gradient are ctn_rows_gemm, weight and bias gradient one ctn_rows_wgrad over all rows."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ops._require_cuda(x, weight)
        shape = x.shape
        x2 = x.detach().reshape(-1, shape[-1]).float()
        w = weight.detach().float()
        ctx.saved = (x2, w, bias is not None, shape)
        y = ops.rows_gemm(x2, w, None if bias is None else bias.detach().float().contiguous())
        return y.view(shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, w, has_bias, shape = ctx.saved
        dy2 = dy.reshape(-1, w.shape[0]).float()
        dx = ops.rows_gemm(dy2, w, w_is_km=True).view(shape) if ctx.needs_input_grad[0] else None
        db = torch.zeros((w.shape[0],), device=dy2.device, dtype=torch.float32) if has_bias else None
        dW = ops.rows_wgrad(dy2, x2, dbias=db)
        return dx, dW, db


class LstmFn(torch.autograd.Function):
    """nn.LSTM(batch_first=True, num_layers=1), one or two directions, zero initial state: x (N, T, F) -> (N, T, D*H) and
    the final states.  The input projection of all steps is one ctn_rows_gemm in front of the recurrence, the
    recurrence itself and its backward pass through time are ctn_lstm_fwd / ctn_lstm_bwd (one launch per walk), and
    the parameter / input gradients are row products over all (sequence, step) pairs behind it (ctn_rows_wgrad /
    ctn_rows_gemm): no launch of a GEMM library anywhere.
    params: weight_ih, weight_hh, bias_ih, bias_hh per direction, in nn.LSTM's _flat_weights order."""

    @staticmethod
    def forward(ctx, x, n_dir, *params):
        ops._require_cuda(x)
        N, T, F_in = x.shape
        w_ih = [params[4 * d].detach() for d in range(n_dir)]
        w_hh = [params[4 * d + 1].detach() for d in range(n_dir)]
        bias = [(params[4 * d + 2] + params[4 * d + 3]).detach() for d in range(n_dir)]
        H = w_hh[0].shape[1]
        x2 = x.detach().reshape(N * T, F_in).float()
        w_ih_all = torch.cat(w_ih, dim=0)                                  # (D*4H, F)
        xproj = ops.rows_gemm(x2, w_ih_all, torch.cat(bias)).view(N, T, n_dir, 4 * H)
        whh = torch.stack(w_hh).contiguous()                              # (D, 4H, H)
        hout, save = ops.lstm_fwd(xproj, whh)
        ctx.saved = (x2, w_ih_all, whh, hout, save, n_dir, (N, T, F_in, H))
        # final states as nn.LSTM returns them, (D, N, H): the forward direction's last step, the reverse direction's
        # first (the step it consumes last); the cell state is the fifth saved plane.  Not differentiated.
        last = [T - 1, 0][:n_dir]
        h_n = torch.stack([hout[:, t, d] for d, t in enumerate(last)])
        c_n = torch.stack([save[:, t, d, 4] for d, t in enumerate(last)])
        ctx.mark_non_differentiable(h_n, c_n)
        return hout.view(N, T, n_dir * H), h_n, c_n

    @staticmethod
    def backward(ctx, dout, _dh_n=None, _dc_n=None):
        x2, w_ih_all, whh, hout, save, n_dir, (N, T, F_in, H) = ctx.saved
        dgates, db_all = ops.lstm_bwd(dout.contiguous().float().view(N, T, n_dir, H), save, whh)      # (N, T, D, 4H), (D, 4H)
        db_all = db_all.view(-1)
        dg2 = dgates.view(N * T, n_dir * 4 * H)
        dx = ops.rows_gemm(dg2, w_ih_all, w_is_km=True).view(N, T, F_in) if ctx.needs_input_grad[0] else None
        dw_ih_all = ops.rows_wgrad(dg2, x2)                               # (D*4H, F)
        # recurrent weights: sum over (n, t) of dgates[n, t]^T h[n, t -+ 1], h of the step before t in the direction's own
        # order.  Over the flat (n, t) index that is one shifted product of the whole tensors, minus the N - 1 terms that
        # pair the first step of a sequence with the last state of its neighbour (no shifted copy of h is made): both are
        # row products over strided views, the second one subtracted into the same tensor.
        dg3 = dgates.view(N * T, n_dir, 4 * H)
        h3 = hout.view(N * T, n_dir, H)
        grads = []
        for d in range(n_dir):
            dg, h = dg3[:, d], h3[:, d]                                   # (N*T, 4H), (N*T, H): strided views
            dw_hh = torch.zeros((4 * H, H), device=hout.device, dtype=torch.float32)
            if N * T >= 2:
                if d == 0:
                    ops.rows_wgrad(dg[1:], h[:-1], dW=dw_hh)
                    if N > 1:
                        ops.rows_wgrad(dgates[1:, 0, d], hout[:-1, T - 1, d], dW=dw_hh, sign=-1.0)
                else:
                    ops.rows_wgrad(dg[:-1], h[1:], dW=dw_hh)
                    if N > 1:
                        ops.rows_wgrad(dgates[:-1, T - 1, d], hout[1:, 0, d], dW=dw_hh, sign=-1.0)
            db = db_all[d * 4 * H:(d + 1) * 4 * H]
            grads += [dw_ih_all[d * 4 * H:(d + 1) * 4 * H], dw_hh, db, db]
        return (dx, None) + tuple(grads)


# ------------------------------------------------------------------------------------------------ axis shuffles of the dual-path blocks
class RowsFromBlocksFn(torch.autograd.Function):
    """(B, F, S, K) -> the recurrence's rows (B*S, K, F) (walk_chunks False) or (B*K, S, F) (True), i.e.
    input.permute(0, 2, 3, 1) / permute(0, 3, 2, 1) made contiguous (src/models/dprnn.py:83-84, :135-136), as one
    tiled transpose; the gradient takes the inverse one and comes back contiguous."""

    @staticmethod
    def forward(ctx, x, walk_chunks):
        ops._require_cuda(x)
        ctx.walk = walk_chunks
        r = ops.rows_from_blocks(x.detach().float().contiguous(), walk_chunks)
        ctx.rshape = r.shape
        return r.view(r.shape[0] * r.shape[1], r.shape[2], r.shape[3])

    @staticmethod
    def backward(ctx, dr):
        return ops.blocks_from_rows(dr.contiguous().float().view(ctx.rshape), ctx.walk), None


class FeatureFirstFn(torch.autograd.Function):
    """(B, L, F) -> (B, F, L) (src/models/dprnn.py:88, :140), written in the layout the norm kernels read."""

    @staticmethod
    def forward(ctx, r):
        ops._require_cuda(r)
        return ops.feature_first(r.detach().float().contiguous())

    @staticmethod
    def backward(ctx, g):
        g = g.float()
        return ops.feature_last(g if g.stride(2) == 1 else g.contiguous())


class SwapAddFn(torch.autograd.Function):
    """z (B, F, K, S), x (B, F, S, K) -> z.permute(0, 1, 3, 2) + x (src/models/dprnn.py:144-146) in one pass."""

    @staticmethod
    def forward(ctx, z, x):
        ops._require_cuda(z, x)
        z = z.detach().float()
        if z.stride(3) != 1 or z.stride(2) != z.shape[3]:
            z = z.contiguous()
        return ops.swap_blocks(z, addend=x.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous().float()
        dz = ops.swap_blocks(dout, act=True) if ctx.needs_input_grad[0] else None
        return dz, dout if ctx.needs_input_grad[1] else None


# ------------------------------------------------------------------------------------------------ chunking
class SegmentFn(torch.autograd.Function):
    """Segment1d.forward (src/models/transform.py:15-31) of the input zero-padded by (pad_left, as far as S chunks reach):
    (B, C, T) -> (B, C, S, K).  Backward is the overlap-add of the gradient, trimmed to T."""

    @staticmethod
    def forward(ctx, x, chunk_size, hop_size, S, pad_left):
        x = as_act(x.detach().float())
        ctx.meta = (hop_size, x.shape[2], pad_left)
        return ops.segment(x, chunk_size, hop_size, S, pad_left)

    @staticmethod
    def backward(ctx, dy):
        hop_size, T, pad_left = ctx.meta
        return ops.overlap_add(dy.contiguous().float(), hop_size, T, pad_left), None, None, None, None


class OverlapAddFn(torch.autograd.Function):
    """OverlapAdd1d.forward (src/models/transform.py:45-61) followed by a trim of pad_left frames in front and of
    everything beyond T frames: (B, C, S, K) -> (B, C, T).  Backward is the segmentation of the gradient."""

    @staticmethod
    def forward(ctx, y, hop_size, T, pad_left):
        y = y.detach().contiguous().float()
        ctx.meta = (hop_size, y.shape[2], y.shape[3], pad_left)
        return ops.overlap_add(y, hop_size, T, pad_left)

    @staticmethod
    def backward(ctx, dx):
        hop_size, S, K, pad_left = ctx.meta
        return ops.segment(as_act(dx.float()), K, hop_size, S, pad_left), None, None, None


# ------------------------------------------------------------------------------------------------ pointwise conv
class Conv1x1Fn(torch.autograd.Function):
    """nn.Conv1d(C_in, C_out, kernel_size=1) forward/backward on the MFMA GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = as_act(x.detach())
        B, K, T = x.shape
        M = weight.shape[0]
        w2 = weight.detach().view(M, K)
        Wp = _prepared(None, None, K, M, [(w2, True, 0, 0)], x.device)
        Wt = None
        if Wp is None:
            Wt = torch.empty((K, M), device=x.device, dtype=torch.float32)
            ops.transpose(w2, Wt, M)
        y = new_act(B, M, T, x.device)
        ops.gemm_nn(M, T, K, B, Wt, M, x, y, bias=None if bias is None else bias.detach(), A_prep=Wp)
        ctx.saved = (x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved
        B, K, T = x.shape
        M = weight.shape[0]
        dy = as_act(dy)
        dx = new_act(B, K, T, x.device)
        w2 = weight.detach().view(M, K)
        ops.gemm_nn(K, T, M, B, w2, K, dy, dx, A_prep=_prepared(None, None, M, K, [(w2, False, 0, 0)], x.device))
        dW = torch.zeros((M, K), device=x.device, dtype=torch.float32)
        db = None if bias is None else torch.zeros((M,), device=x.device, dtype=torch.float32)
        ops.gemm_nt(M, K, T, B, dy, x, dW, K, a_rowsum=db)
        return dx, dW.view(weight.shape), db


class DepthwiseFn(torch.autograd.Function):
    """Plain depthwise nn.Conv1d(C, C, P, stride=1, dilation=d, groups=C) with explicit zero padding
    (src/modules/conv.py:22,26)."""

    @staticmethod
    def forward(ctx, x, weight, bias, dilation, pad_left, pad_right):
        x = as_act(x.detach())
        B, Cc, T = x.shape
        P = weight.shape[-1]
        Tout = T + pad_left + pad_right - (P - 1) * dilation
        if Tout != T:
            raise NotImplementedError("depthwise kernel supports 'same'-length output only "
                                      "(pad_left + pad_right == (P-1)*dilation)")
        y = new_act(B, Cc, T, x.device)
        ops.dw_fwd(x, weight.detach().view(Cc, P), None if bias is None else bias.detach(), y, dilation, pad_left)
        ctx.saved = (x, weight, bias, dilation, pad_left)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, dilation, pad_left = ctx.saved
        B, Cc, T = x.shape
        P = weight.shape[-1]
        dy = as_act(dy)
        dw = torch.zeros((Cc, P), device=x.device, dtype=torch.float32)
        db = None if bias is None else torch.zeros((Cc,), device=x.device, dtype=torch.float32)
        dx = new_act(B, Cc, T, x.device)
        ops.dw_bwd(dy, x, weight.detach().view(Cc, P), dw, db, dx, dilation, pad_left)
        return dx, dw.view(weight.shape), db, None, None, None


class DepthwiseGenericFn(torch.autograd.Function):
    """Depthwise nn.Conv1d(C, C, K, stride, padding, dilation, groups=C) of any geometry (src/modules/conv.py:22,26)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, dilation, padding):
        x = as_act(x.detach().float())
        B, Cc, T = x.shape
        K = weight.shape[-1]
        Tout = (T + 2 * padding - dilation * (K - 1) - 1) // stride + 1
        if Tout < 1:
            raise RuntimeError("Calculated padded input size per channel: ({}). Kernel size: ({}). Kernel size can't be "
                               "greater than actual input size".format(T + 2 * padding, dilation * (K - 1) + 1))
        w = weight.detach().reshape(Cc, K).contiguous().float()
        y = ops.dwconv_generic_fwd(x, w, None if bias is None else bias.detach().float(), stride, dilation, padding, Tout)
        ctx.saved = (x, w, weight.shape, bias is not None, stride, dilation, padding)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, wshape, has_bias, stride, dilation, padding = ctx.saved
        dx, dW, db = ops.dwconv_generic_bwd(x, w, as_act(dy), stride, dilation, padding, ctx.needs_input_grad[0],
                                            has_bias)
        return dx, dW.view(wshape), db, None, None, None


# ------------------------------------------------------------------------------------------------ encoder / decoder
def _enc_geometry(T, L, S):
    padding = (S - (T - L) % S) % S                   # src/models/conv_tasnet.py:145
    pad_left = padding // 2
    Tf = (T + padding - L) // S + 1
    return padding, pad_left, Tf


class EncoderFn(torch.autograd.Function):
    """Encoder.forward (src/models/filterbank.py:222-230) on an already padded (B, 1, T) input."""

    @staticmethod
    def forward(ctx, x, weight, stride, relu):
        ops._require_cuda(x, weight)
        B, C_in, T = x.shape
        if C_in != 1:
            raise NotImplementedError("HIP encoder supports in_channels == 1")
        N, _, L = weight.shape
        Tf = (T - L) // stride + 1
        x2 = x.detach().reshape(B, T).contiguous().float()
        w = new_act(B, N, Tf, x.device)
        ops.encoder_fwd(x2, 0, weight.detach(), stride, relu, w)
        ctx.saved = (x2, weight, stride, relu, w)
        ctx.set_materialize_grads(False)
        return w

    @staticmethod
    def backward(ctx, dw):
        x2, weight, stride, relu, w = ctx.saved
        if dw is None:
            return None, None, None, None
        N, _, L = weight.shape
        dW = torch.zeros((N, L), device=w.device, dtype=torch.float32)
        dwt = new_act(w.shape[0], N, w.shape[2], w.device) if ctx.needs_input_grad[0] else None
        ops.encoder_bwd(x2, 0, N, L, stride, relu, w, dW, dw_a=as_act(dw), dwt_out=dwt)
        dx = None
        if dwt is not None:   # transposed convolution of dwt with the same basis
            B, T = x2.shape
            dx = torch.zeros((B, 1, T), device=w.device, dtype=torch.float32)
            ops.maskdecode_fwd(dwt, None, 1, weight.detach(), stride, dx, 0)
        return dx, dW.view(weight.shape), None, None


class DecoderFn(torch.autograd.Function):
    """Decoder.forward (src/models/filterbank.py:245-247): ConvTranspose1d(N, 1, L, stride, bias=False)."""

    @staticmethod
    def forward(ctx, w, weight, stride):
        w = as_act(w.detach())
        B, N, Tf = w.shape
        L = weight.shape[-1]
        Tout = (Tf - 1) * stride + L
        out = torch.empty((B, 1, Tout), device=w.device, dtype=torch.float32)
        ops.maskdecode_fwd(w, None, 1, weight.detach(), stride, out, 0)
        ctx.saved = (w, weight, stride)
        return out

    @staticmethod
    def backward(ctx, dout):
        w, weight, stride = ctx.saved
        B, N, Tf = w.shape
        L = weight.shape[-1]
        dout = dout.contiguous().float()
        dWd = torch.zeros((N, L), device=w.device, dtype=torch.float32)
        dw = new_act(B, N, Tf, w.device)
        ops.maskdecode_bwd(dout, 0, w, None, 1, weight.detach(), stride, False, None, dw, dWd)
        return dw, dWd.view(weight.shape), None


# ------------------------------------------------------------------------------------------------ whole model
class ModelParams:
    """Parameter tensors of a ConvTasNet in a fixed order for ConvTasNetFn.

    order: enc_W, n0_gamma, n0_beta, Wb, bb, [layer params ...], a_sep, Wm, bm, dec_W"""

    def __init__(self, tensors, cfg):
        self.cfg = cfg
        R, X = cfg["sep_num_blocks"], cfg["sep_num_layers"]
        it = iter(tensors)
        self.tensors = list(tensors)
        self.enc_W, self.g0, self.be0, self.Wb, self.bb = (next(it) for _ in range(5))
        self.layers = []
        for r in range(R):
            for l in range(X):
                dual = not (r == R - 1 and l == X - 1)
                n = len(Layer.NAMES_DUAL if dual else Layer.NAMES_SINGLE)
                self.layers.append((Layer([next(it) for _ in range(n)], dual), 2 ** l))
        self.a_sep, self.Wm, self.bm, self.dec_W = (next(it) for _ in range(4))
        self.planes = None
        store = cfg.get("weight_planes")
        if store is not None:                  # a PreparedWeights over these very parameters, refreshed this step
            self.planes = store.sep
            for (L, _), d in zip(self.layers, store.layers):
                L.planes = d


def _separator_parts(mp):
    """key -> (K, M, K1, parts) of the weight operands of the separator's own GEMMs (bottleneck and mask convolution,
    forward and input gradient)."""
    Bc, N = mp.Wb.shape[0], mp.Wb.shape[1]
    Mm, Sc = mp.Wm.shape[0], mp.Wm.shape[1]
    Wb, Wm = mp.Wb.detach().view(Bc, N), mp.Wm.detach().view(Mm, Sc)
    return {"Wbf": (N, Bc, None, [(Wb, True, 0, 0)]), "Wbd": (Bc, N, None, [(Wb, False, 0, 0)]),
            "Wmf": (Sc, Mm, None, [(Wm, True, 0, 0)]), "Wmd": (Mm, Sc, None, [(Wm, False, 0, 0)])}


def separator_forward(mp, w, st0_acc, eps, sigmoid=True):
    """norm -> bottleneck -> TCN -> PReLU -> mask conv -> sigmoid (src/models/conv_tasnet.py:359-378).
    w act (B, N, Tf) with its norm statistics already accumulated in st0_acc.  Returns mask, saved."""
    cfg = mp.cfg
    causal = cfg["causal"]
    B, N, T = w.shape
    dev = w.device
    Bc = mp.Wb.shape[0]
    st0 = ops.finalize_stats(st0_acc, B, N, T, eps, dev)
    sparts = _separator_parts(mp)
    Wbp = _prepared(mp.planes, "Wbf", *sparts["Wbf"][:2], sparts["Wbf"][3], dev)
    Wbt = None
    if Wbp is None:
        Wbt = mp.planes.get("Wbt") if mp.planes is not None else None
        if Wbt is None:
            Wbt = torch.empty((N, Bc), device=dev, dtype=torch.float32)
            ops.transpose(mp.Wb.detach().view(Bc, N), Wbt, Bc)
    x = new_act(B, Bc, T, dev)
    ops.gemm_nn(Bc, T, N, B, Wbt, Bc, w, x, xflags=XF_NORM, stat=st0, gamma=mp.g0.detach().view(-1),
                beta=mp.be0.detach().view(-1), bias=mp.bb.detach(), A_prep=Wbp)
    Sc = mp.layers[0][0].Sc
    skip = new_act(B, Sc, T, dev)
    saved_layers = []
    for i, (L, dil) in enumerate(mp.layers):
        x_next, sv = layer_forward(L, x, skip, i > 0, dil, causal, eps)
        saved_layers.append(sv)
        x = x_next
    M = mp.Wm.shape[0]
    Wmp = _prepared(mp.planes, "Wmf", *sparts["Wmf"][:2], sparts["Wmf"][3], dev)
    Wmt = None
    if Wmp is None:
        Wmt = mp.planes.get("Wmt") if mp.planes is not None else None
        if Wmt is None:
            Wmt = torch.empty((Sc, M), device=dev, dtype=torch.float32)
            ops.transpose(mp.Wm.detach().view(M, Sc), Wmt, M)
    mask = new_act(B, M, T, dev)
    ops.gemm_nn(M, T, Sc, B, Wmt, M, skip, mask, xflags=XF_PRELU, alpha=mp.a_sep.detach(), bias=mp.bm.detach(),
                emode=EPI_SIGMOID if sigmoid else EPI_STORE, A_prep=Wmp)
    return mask, (st0, skip, saved_layers)


def separator_backward(mp, w, saved, dz, sink, eps):
    """dz act (B, n_src*N, T): gradient w.r.t. the mask logits.  Returns dn0 (gradient w.r.t. the normalised
    separator input, before the entry norm's backward) and the entry-norm Stat with c1/c2 filled."""
    cfg = mp.cfg
    causal = cfg["causal"]
    st0, skip, saved_layers = saved
    B, N, T = w.shape
    dev = w.device
    Bc = mp.Wb.shape[0]
    Sc = mp.layers[0][0].Sc
    M = mp.Wm.shape[0]
    # mask conv: weight/bias gradient, input gradient through the separator PReLU
    ops.gemm_nt(M, Sc, T, B, dz, skip, sink.g(mp.Wm).view(M, Sc), Sc, a_rowsum=sink.g(mp.bm), bflags=XF_PRELU,
                b_alpha=mp.a_sep.detach())
    dskip = new_act(B, Sc, T, dev)
    sparts = _separator_parts(mp)
    ops.gemm_nn(Sc, T, M, B, mp.Wm.detach().view(M, Sc), Sc, dz, dskip, emode=EPI_PRELUBWD, E=skip,
                ealpha=mp.a_sep.detach(), esums=sink.alpha_slot(mp.a_sep),
                A_prep=_prepared(mp.planes, "Wmd", *sparts["Wmd"][:2], sparts["Wmd"][3], dev))
    dout = None
    # cfg["region_hook"](layer index): called after the backward of layer `index` has been enqueued, for the indices in
    # cfg["region_layers"] -- the first layer of every TCN block but block 0.  Everything behind that layer in the
    # parameter order is final then (the trainer starts that region's gradient exchange while the rest of the backward
    # pass runs).
    hook, hook_layers = cfg.get("region_hook"), cfg.get("region_layers", ())
    for i in range(len(mp.layers) - 1, -1, -1):
        (L, dil), sv = mp.layers[i], saved_layers[i]
        dout = layer_backward(L, sv, dout, dskip, sink, dil, causal, eps)
        if hook is not None and i in hook_layers:
            hook(i, sink)
    # separator bottleneck
    dn0 = new_act(B, N, T, dev)
    bacc0 = Accum(B, T, causal, dev)
    fin0 = ops.coefs_request(bacc0, st0, B, N, T)
    ops.gemm_nn(N, T, Bc, B, mp.Wb.detach().view(Bc, N), N, dout, dn0, emode=EPI_NORMBWD, E=w, ealpha=None,
                estat=st0, egamma=mp.g0.detach().view(-1), eacc=bacc0, fin=fin0,
                A_prep=_prepared(mp.planes, "Wbd", *sparts["Wbd"][:2], sparts["Wbd"][3], dev))
    ops.gemm_nt(Bc, N, T, B, dout, w, sink.g(mp.Wb).view(Bc, N), N, a_rowsum=sink.g(mp.bb), bflags=XF_NORM,
                b_stat=st0, b_gamma=mp.g0.detach().view(-1), b_beta=mp.be0.detach().view(-1))
    if fin0 is None:
        ops.bwd_coefs(bacc0, st0, B, N, T, eps)
    return dn0, st0


class ConvTasNetFn(torch.autograd.Function):
    """ConvTasNet.forward for trainable encoder/decoder, separable dilated TCN with PReLU + norm and a
    sigmoid mask (src/models/conv_tasnet.py:116-171)."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        ops._require_cuda(x, params[0])
        mp = ModelParams(params, cfg)
        B, C_in, T = x.shape
        L, S = cfg["kernel_size"], cfg["stride"]
        N, n_src = cfg["n_basis"], cfg["n_sources"]
        eps, causal = cfg["eps"], cfg["causal"]
        relu = cfg.get("enc_nonlinear") == "relu"
        dev = x.device
        padding, pad_left, Tf = _enc_geometry(T, L, S)
        x2 = x.detach().reshape(B, T).contiguous().float()
        w = new_act(B, N, Tf, dev)
        with ops.zero_arena(dev, (2 * len(mp.layers) + 4) * ops.accum_bytes(B, Tf, causal)):
            acc0 = Accum(B, Tf, causal, dev)
            ops.encoder_fwd(x2, pad_left, mp.enc_W.detach(), S, relu, w, acc0)
            mask, saved = separator_forward(mp, w, acc0, eps)
        out = torch.empty((B, n_src, T), device=dev, dtype=torch.float32)
        ops.maskdecode_fwd(w, mask, n_src, mp.dec_W.detach(), S, out, pad_left)
        ctx.mp, ctx.saved_acts = mp, (x2, w, mask, saved, pad_left, relu)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("the fused Conv-TasNet schedule has no gradient with respect to the mixture; "
                                      "use ConvTasNet.extract_latent (module by module) for that")
        mp = ctx.mp
        cfg = mp.cfg
        x2, w, mask, saved, pad_left, relu = ctx.saved_acts
        B, N, Tf = w.shape
        L, S, n_src, eps = cfg["kernel_size"], cfg["stride"], cfg["n_sources"], cfg["eps"]
        dev = w.device
        sink = GradSink(mp.tensors, cfg.get("grad_flat"), cfg.get("grad_zero", True))
        dout = dout.contiguous().float()
        dz = new_act(B, n_src * N, Tf, dev)
        dw_dec = new_act(B, N, Tf, dev)
        ops.maskdecode_bwd(dout, pad_left, w, mask, n_src, mp.dec_W.detach(), S, True, dz, dw_dec,
                           sink.g(mp.dec_W).view(N, L))
        # (+ per layer the per-sample row sums and dot sums of the heads' weight-gradient launch, HEADS_COEFS_FROM_WGRAD)
        heads_extra = sum((B * (Lr.Bc + Lr.Sc) * 4 + 255) // 256 * 256 + (B * ops._lib.STAT_SLOTS * 8 + 255) // 256 * 256 +
                          ((B + 1) * 8 + 255) // 256 * 256
                          for Lr, _ in mp.layers)
        with ops.zero_arena(dev, (2 * len(mp.layers) + 4) * ops.accum_bytes(B, Tf, cfg["causal"]) + heads_extra):
            dn0, st0 = separator_backward(mp, w, saved, dz, sink, eps)
        ops.encoder_bwd(x2, pad_left, N, L, S, relu, w, sink.g(mp.enc_W).view(N, L), dw_a=dw_dec, dn=dn0, stat=st0,
                        gamma=mp.g0.detach().view(-1), dgamma=sink.g(mp.g0).view(-1), dbeta=sink.g(mp.be0).view(-1))
        grads = sink.finish()
        if cfg.get("grad_detached"):           # the caller reads its flat buffer; autograd must not add views of it
            return (None, None) + (None,) * len(grads)
        return (None, None) + tuple(grads)


class SeparatorFn(torch.autograd.Function):
    """Separator.forward up to the mask logits (src/models/conv_tasnet.py:370-374) on its own:
    input (B, N, T') -> logits (B, n_src*N, T').  The mask non-linearity is applied by the caller."""

    @staticmethod
    def forward(ctx, w, cfg, *params):
        # params: g0, be0, Wb, bb, layers..., a_sep, Wm, bm  (ModelParams order without encoder/decoder)
        w = as_act(w.detach())
        B, N, T = w.shape
        mp = ModelParams((None,) + tuple(params) + (None,), cfg)
        mp.tensors = list(params)
        acc0 = Accum(B, T, cfg["causal"], w.device)
        ops.norm_stats(w, None, acc0)
        logits, saved = separator_forward(mp, w, acc0, cfg["eps"], sigmoid=False)
        ctx.mp, ctx.saved_acts = mp, (w, saved)
        return logits

    @staticmethod
    def backward(ctx, dz):
        mp = ctx.mp
        cfg = mp.cfg
        w, saved = ctx.saved_acts
        B, N, T = w.shape
        eps = cfg["eps"]
        sink = GradSink(mp.tensors)
        dn0, st0 = separator_backward(mp, w, saved, as_act(dz), sink, eps)
        g0 = mp.g0.detach().view(-1)
        scratch = Accum(B, T, cfg["causal"], w.device)
        ops.norm_bwd_reduce(dn0, w, None, st0, g0, scratch, sink.g(mp.g0).view(-1), sink.g(mp.be0).view(-1))
        dw = new_act(B, N, T, w.device)
        ops.norm_bwd_apply(dn0, w, None, st0, g0, dw, None)
        grads = sink.finish()
        return (dw, None) + tuple(grads)
