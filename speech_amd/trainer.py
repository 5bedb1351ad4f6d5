"""Data-parallel training step for Conv-TasNet on MI355X: one process per GPU, flat fp32 parameter /
gradient / Adam-state buffers, gradient all-reduce on RCCL, fused clip + Adam kernel.

Reproduces the update of the reference's loop (egs/wsj0-mix/common/src/driver.py:141-157 with the
setup of egs/wsj0-mix/conv-tasnet/local/train.py:92-118): forward -> PIT SI-SDR (mean over the
GLOBAL batch) -> backward -> clip_grad_norm_(max_norm) -> Adam(lr).  The reference reaches several
GPUs through single-process nn.DataParallel; here every rank owns an equal shard of the batch, so
mean-of-shard-means followed by gradient averaging is the same global-batch mean, and clipping runs
after the all-reduce on the averaged gradient.
"""
import torch
import torch.distributed as dist

from . import ops
from . import functional
from .functional import ConvTasNetFn, ModelParams, PreparedWeights, flat_layout


def allreduce_flat(flat, n_buckets=1, group=None):
    """Sum-all-reduce a flat gradient buffer in `n_buckets` contiguous chunks (asynchronously, then wait).
    The result is independent of n_buckets; with the NCCL backend this is RCCL over xGMI."""
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal / debugging without RCCL (e.g. several ranks on one GPU): reduce a host copy
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return flat
    n = flat.numel()
    n_buckets = max(1, min(n_buckets, n))
    per = (n + n_buckets - 1) // n_buckets
    works = []
    for i in range(n_buckets):
        chunk = flat[i * per:min((i + 1) * per, n)]
        if chunk.numel():
            works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    return flat


def shard_batch(mixture, sources, rank, world):
    """Equal contiguous shard of a global batch for `rank` (the batch must divide evenly)."""
    B = mixture.shape[0]
    if B % world:
        raise ValueError("global batch {} does not divide over {} ranks".format(B, world))
    per = B // world
    return mixture[rank * per:(rank + 1) * per], sources[rank * per:(rank + 1) * per]


class FlatTrainer:
    def __init__(self, model, criterion, lr=1e-3, max_norm=5.0, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 n_buckets=1, n_streams=1, use_graph=False, overlap_exchange=None):
        self.model, self.criterion = model, criterion
        self.lr, self.max_norm, self.betas, self.eps = lr, max_norm, betas, eps
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.n_buckets = max(1, n_buckets)
        # n_streams > 1: the rank's shard is cut into that many sub-batches whose forward/backward passes run on
        # separate HIP streams.  Every kernel launch of one pass fills the chip with blocks that move through
        # their load, MFMA and store phases in lockstep; two passes in flight put one's stores beside the
        # other's MFMAs.  The result is the same sum of per-utterance gradients (order of additions aside).
        self.n_streams = max(1, n_streams)
        self.streams = None
        # use_graph: after two eager steps the forward/backward pass (about 570 kernel launches) is captured in a
        # HIP graph per input shape and replayed from static input buffers; the all-reduce and the optimizer
        # (whose bias correction changes every step) stay outside the graph.
        self.use_graph = bool(use_graph)
        self._graphs = {}
        self.max_graphs = 2
        self._eager_steps = 0
        # overlap_exchange: the gradient buffer is all-reduced region by region while the backward pass is still
        # running -- one region per TCN block, in the order the backward pass completes them (last block first) -- on
        # the process group's own stream, instead of in one piece after the pass.  It needs the collectives between
        # the kernel launches, so it applies to eager launches; with use_graph the whole pass is one graph replay and
        # the exchange follows it (n_buckets pieces in flight together).  Opt-in (True): the default, None or False,
        # is the exchange after the pass.  The overlapped form issues asynchronous collectives from inside the
        # autograd pass on a side stream; its host-copy (gloo) branch is exercised by the two-rank tests, its RCCL
        # branch has not run on more than one GPU, and a wrong stream dependency there would clip and step on partly
        # reduced gradients without any error -- so it stays off until a multi-GPU run has held it to the
        # single-process update.
        self.overlap_exchange = overlap_exchange
        self._pending = []
        if not model._fused_ok() or model.dec_basis != 'trainable':
            raise NotImplementedError("FlatTrainer drives the fused Conv-TasNet schedule with a trainable encoder / "
                                      "decoder pair only")
        params = model.fused_parameters()
        others = set(id(p) for p in model.parameters()) - set(id(p) for p in params)
        if others:
            raise RuntimeError("model has parameters outside the fused schedule")
        dev = params[0].device
        if dev.type != "cuda":
            raise ops._lib.CtnError("FlatTrainer needs the model on a GPU")
        offsets, total = flat_layout(params)       # the gradient sink lays its buffer out the same way
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(params, offsets):
                view = self.flat_p[off:off + p.numel()].view(p.shape)
                view.copy_(p)
                p.data = view                      # parameters become views of the flat buffer
        self.params, self.offsets = params, offsets
        self.flat_g = torch.zeros_like(self.flat_p)
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self.step_count = 0
        # weight operands of the pointwise GEMMs laid out once per pass (k-major copies; prepared planes on request)
        self._prepared = PreparedWeights(ModelParams(self.params, model.fused_config()),
                                         planes=functional.USE_PREPARED_WEIGHTS)
        if self.world > 1:                         # identical replicas: broadcast rank 0's weights
            dist.broadcast(self.flat_p, src=0, group=self.group)

    def moment_views(self):
        """{id(parameter): (exp_avg, exp_avg_sq)} as views of the flat Adam buffers, shaped like the parameter."""
        return {id(p): (self.m[off:off + p.numel()].view(p.shape), self.v[off:off + p.numel()].view(p.shape))
                for p, off in zip(self.params, self.offsets)}

    def forward_backward(self, mixture, sources):
        """Fills self.flat_g with this rank's gradient of the shard-mean loss; returns (loss, perm).
        The gradients live in flat_g only (parameters' .grad stay None)."""
        if not self.use_graph:
            return self._forward_backward(mixture, sources)
        key = (tuple(mixture.shape), tuple(sources.shape))
        entry = self._graphs.get(key)
        if entry is None and len(self._graphs) >= self.max_graphs:
            # every captured shape keeps its own pool of activations (several GB at batch 32): other shapes -- a ragged
            # last batch, variable-length training -- run eagerly.  Use fixed-length batches (drop_last) with use_graph.
            return self._forward_backward(mixture, sources)
        if entry is None:
            if self._eager_steps < 2:              # lazy initialisation (library load, caches) happens eagerly
                self._eager_steps += 1
                return self._forward_backward(mixture, sources)
            static_mix, static_src = mixture.clone(), sources.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                # thread_local: a process group's watchdog thread may touch the device while this thread captures
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    loss, perm = self._forward_backward(static_mix, static_src)
            except RuntimeError as exc:            # capture refused: keep training with eager launches
                import sys
                print("FlatTrainer: HIP graph capture failed ({}); continuing with eager launches".format(exc),
                      file=sys.stderr)
                self.use_graph = False
                torch.cuda.synchronize()
                return self._forward_backward(mixture, sources)
            entry = (graph, static_mix, static_src, loss, perm)
            self._graphs[key] = entry
        graph, static_mix, static_src, loss, perm = entry
        static_mix.copy_(mixture)
        static_src.copy_(sources)
        graph.replay()
        return loss.clone(), perm.clone()

    # ------------------------------------------------------------------ regions of the flat gradient buffer
    def _region_table(self):
        """[(first layer index or None, start, end)] in the order the backward pass completes them: the parameters of
        TCN block r and everything behind them that is not in an earlier region.  Region 0 is the last block plus the
        mask convolution and the decoder; the last region is block 0 plus the encoder, entry norm and bottleneck and is
        complete only when the whole pass is."""
        if getattr(self, "_regions", None) is not None:
            return self._regions
        cfg = self.model.fused_config()
        R, X = cfg["sep_num_blocks"], cfg["sep_num_layers"]
        n_dual, n_single = 14, 12                      # parameters per residual layer (functional.Layer.NAMES_*)
        assert len(self.params) == 5 + (R * X - 1) * n_dual + n_single + 4
        total = self.flat_g.numel()
        regions, end = [], total
        for r in range(R - 1, 0, -1):
            first = 5 + r * X * n_dual                # parameter index of block r's first layer
            regions.append((r * X, self.offsets[first], end, first))
            end = self.offsets[first]
        regions.append((None, 0, end, 0))
        self._regions = regions
        return regions

    def _overlapping(self):
        if self.world == 1:
            return False
        return bool(self.overlap_exchange) and not self.use_graph

    def _exchange_region(self, start, end, ready_streams):
        """All-reduce flat_g[start:end] once every stream in `ready_streams` has reached this point of its work."""
        chunk = self.flat_g[start:end]
        if self._prep is None:
            self._prep = torch.cuda.Stream(device=self.flat_p.device)
        for st in ready_streams:
            self._prep.wait_stream(st)
        if dist.get_backend(self.group) == "gloo":     # rehearsal without RCCL: blocking, through the host
            self._prep.synchronize()
            host = chunk.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            with torch.cuda.stream(self._prep):
                chunk.copy_(host)
            self._pending.append(self._prep.record_event())
            return
        with torch.cuda.stream(self._prep):            # the collective's stream starts behind _prep, i.e. behind the streams
            self._pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _finish_exchange(self):
        """The current stream waits for every region's collective."""
        for w in self._pending:
            w.wait()
        self._pending = []

    _prep = None

    def _forward_backward(self, mixture, sources):
        B = mixture.shape[0]
        n = min(self.n_streams, B)
        for p in self.params:
            p.grad = None
        overlap = self._overlapping()
        regions = self._region_table() if overlap else []
        hook_layers = {first: k for k, (first, _, _, _) in enumerate(regions) if first is not None}
        self._exchanged = False
        if not self._prepared.planes and functional.USE_PREPARED_WEIGHTS:        # switched on after construction
            self._prepared = PreparedWeights(ModelParams(self.params, self.model.fused_config()), planes=True)
        self._prepared.refresh()                   # on the stream every sub-batch stream waits for below
        planes = self._prepared
        if n == 1:
            main = torch.cuda.current_stream()
            cfg = self.model.fused_config()
            cfg["grad_flat"], cfg["grad_detached"] = self.flat_g, True
            cfg["weight_planes"] = planes
            if overlap:
                def hook(layer, sink):
                    _, start, end, first_param = regions[hook_layers[layer]]
                    sink.fold_alphas(first_param, None)
                    self._exchange_region(start, end, [main])
                cfg["region_hook"], cfg["region_layers"] = hook, set(hook_layers)
            out = ConvTasNetFn.apply(mixture, cfg, *self.params)
            loss, perm = self.criterion(out, sources)
            loss.backward()
            if overlap:
                _, start, end, _ = regions[-1]
                self._exchange_region(start, end, [main])
                self._finish_exchange()
                self._exchanged = True
            return loss.detach(), perm
        if self.streams is None:
            self.streams = [torch.cuda.Stream(device=self.flat_p.device) for _ in range(self.n_streams)]
        main = torch.cuda.current_stream()
        self.flat_g.zero_()
        bounds = [B * i // n for i in range(n + 1)]
        losses, perms = [], []
        for i in range(n):
            st = self.streams[i]
            st.wait_stream(main)                   # inputs and the zeroed gradient buffer are ready
            mixture.record_stream(st)
            sources.record_stream(st)
            lo, hi = bounds[i], bounds[i + 1]
            with torch.cuda.stream(st):
                cfg = self.model.fused_config()
                cfg["grad_flat"], cfg["grad_zero"], cfg["grad_detached"] = self.flat_g, False, True
                cfg["weight_planes"] = planes
                if overlap:
                    # every sub-batch folds its own slope gradients when it closes a region; the region is exchanged
                    # when the LAST sub-batch has closed it (the host enqueues the sub-batches one after the other, so
                    # by then the earlier streams' work on the region is already in their queues)
                    def hook(layer, sink, last=(i == n - 1), done=self.streams[:i + 1]):
                        _, start, end, first_param = regions[hook_layers[layer]]
                        sink.fold_alphas(first_param, None)
                        if last:
                            self._exchange_region(start, end, done)
                    cfg["region_hook"], cfg["region_layers"] = hook, set(hook_layers)
                out = ConvTasNetFn.apply(mixture[lo:hi], cfg, *self.params)
                loss_i, perm_i = self.criterion(out, sources[lo:hi])
                loss_i = loss_i * ((hi - lo) / B)  # mean over the sub-batch -> share of the shard mean
                loss_i.backward()
                losses.append(loss_i.detach())
                perms.append(perm_i)
        for i in range(n):
            main.wait_stream(self.streams[i])
        if overlap:
            _, start, end, _ = regions[-1]
            self._exchange_region(start, end, [main])
            self._finish_exchange()
            self._exchanged = True
        return torch.stack(losses).sum(), torch.cat(perms, dim=0)

    def reduce_gradients(self):
        if self.world == 1 or getattr(self, "_exchanged", False):   # one rank, or exchanged during the backward pass
            self._exchanged = False
            return
        allreduce_flat(self.flat_g, self.n_buckets, self.group)

    def optimizer_step(self):
        self.step_count += 1
        use_clip = bool(self.max_norm)
        if use_clip:
            self.sumsq.zero_()
            ops.sumsq(self.flat_g, self.sumsq)
        ops.adam_step(self.flat_p, self.flat_g, self.m, self.v, self.sumsq if use_clip else None,
                      self.max_norm if use_clip else 0.0, 1.0 / self.world, self.lr, self.betas[0], self.betas[1],
                      self.eps, self.step_count)
        # the kernel wrote the parameters through raw pointers: torch's version counters did not move
        functional.invalidate_prepared_cache()

    def step(self, mixture, sources):
        """One optimizer step; returns the (device) loss of this rank's shard and the permutations."""
        loss, perm = self.forward_backward(mixture, sources)
        self.reduce_gradients()
        self.optimizer_step()
        return loss, perm
