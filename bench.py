"""Benchmark: Conv-TasNet training utterances/s (paper config, 2 speakers, 4 s @ 8 kHz) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward -> PIT SI-SDR -> backward -> gradient all-reduce (N > 1) -> clip(5) -> Adam(1e-3) on one
synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line.  Weak scaling: the per-GPU batch is
fixed, value = utterances all ranks processed / max-over-ranks time.

The line also carries
  roofline        live HIP-event timing of the dominant kernel (the pointwise-conv GEMM, gemm_nn_fast_kernel) over one
                  instrumented step: algorithmic fp32 FLOPs of its launches / their summed duration, against the ceiling
                  of the arithmetic that EXECUTES: in the default mode every fp32 product tile costs six
                  v_mfma_f32_32x32x16_bf16, so the ceiling is the 2.5 PFLOP/s dense bf16 peak / 6 (f32 MFMAs: the 157.3
                  TFLOP/s f32 matrix peak; two pieces: 2.5 PFLOP/s / 3).  frac cannot exceed 1 by construction; the
                  ratio to the f32 matrix peak is a named secondary field;
  roofline_shapes the launches of one residual layer (and the separator's own GEMMs) one by one: measured duration, MFMA
                  floor (FLOPs / that ceiling) and HBM floor (bytes / 8 TB/s; the bytes are the PMC-measured ones of the
                  committed rocprofv3 summary where it has them, the algorithmic ones otherwise);
  roofline_step   the whole step: sum of PMC bytes per step, the rate that is of 8 TB/s, and both floors;
  roofline_hbm    the fused depthwise kernels against the 8 TB/s HBM3E peak (algorithmic bytes: forward 2 x H x T' x 4,
                  backward 3 x H x T' x 4 per utterance and layer), with the rate of a hand-written 16-byte copy kernel
                  measured in the same process as a yardstick;
  cpu_baseline    the CPU oracle (a PyTorch restatement of the reference's step) timed on this host.
Everything behind `value` is informational and guarded: a failure there is reported in the line ("errors"), never
instead of it.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAPER = dict(n_basis=512, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=512,
             sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3, sep_num_blocks=3,
             sep_num_layers=8, mask_nonlinear="sigmoid", causal=False, n_sources=2, eps=1e-12)
T_SAMPLES = 32000          # 4 s at 8 kHz
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
# matrix instructions per algorithmic fp32 product tile in each arithmetic mode (ops.get_gemm_split())
PRODUCTS = {0: 1, 2: 3, 3: 6}


def executed_peak(pieces):
    """TFLOP/s of ALGORITHMIC fp32 work the matrix unit can deliver in this arithmetic mode: the peak of the instruction
    that executes divided by the number of such instructions per product."""
    return PEAK_F32_MFMA_TFLOPS if pieces == 0 else PEAK_BF16_MFMA_TFLOPS / PRODUCTS[pieces]


def build_model(cfg, device):
    from speech_amd.models.conv_tasnet import ConvTasNet
    torch.manual_seed(111)     # the reference recipes' seed (egs/wsj0-mix/conv-tasnet/train.sh:59)
    m = ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis="trainable",
                   dec_basis="trainable", enc_nonlinear=cfg["enc_nonlinear"],
                   sep_hidden_channels=cfg["sep_hidden_channels"],
                   sep_bottleneck_channels=cfg["sep_bottleneck_channels"],
                   sep_skip_channels=cfg["sep_skip_channels"], sep_kernel_size=cfg["sep_kernel_size"],
                   sep_num_blocks=cfg["sep_num_blocks"], sep_num_layers=cfg["sep_num_layers"], dilated=True,
                   separable=True, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear=cfg["mask_nonlinear"],
                   causal=cfg["causal"], n_sources=cfg["n_sources"])
    return m.to(device)


def _rows(t):
    return 0 if t is None else t.shape[1]


def _nn_label(M, T, K, batch, *a, **k):
    """(label, algorithmic HBM bytes) of one ctn_gemm_nn launch: every operand and result tensor once."""
    from speech_amd import ops
    xf, em = k.get("xflags", 0), k.get("emode", ops.EPI_STORE)
    names = {(512, 128, 0, ops.EPI_STATS): "bottleneck_fwd", (256, 512, ops.XF_PRELU | ops.XF_NORM, ops.EPI_HEADS): "heads_fwd",
             (128, 512, ops.XF_PRELU | ops.XF_NORM, ops.EPI_HEADS): "heads_fwd_last_layer",
             (512, 256, 0, ops.EPI_NORMBWD): "heads_dgrad", (512, 128, 0, ops.EPI_NORMBWD): "skip_head_dgrad_or_entry_dgrad",
             (128, 512, ops.XF_NORMBWD, ops.EPI_ADD): "bottleneck_dgrad", (128, 512, ops.XF_NORM, ops.EPI_STORE): "entry_bottleneck_fwd",
             (1024, 128, ops.XF_PRELU, ops.EPI_SIGMOID): "mask_fwd", (128, 1024, 0, ops.EPI_PRELUBWD): "mask_dgrad"}
    label = names.get((M, K, xf, em), "nn_M{}_K{}_x{}_e{}".format(M, K, xf, em))
    rows = K + M                                   # X (and Xb, whose rows are part of K) read, Y (and Y2) written
    rows += K if k.get("X2") is not None else 0    # second operand of the norm-backward transform
    rows += M if k.get("E") is not None else 0     # pre-activation tensor of a norm / PReLU backward epilogue
    if k.get("R") is not None:                     # residual input
        rows += k.get("Mo", 0) or M
    if k.get("accumulate"):                        # skip sum read before it is written
        rows += M - k.get("Mo", 0)
    return label, rows * T * 4.0 * batch


def _nt_label(M, N, T, batch, *a, **k):
    from speech_amd import ops
    af, bf = k.get("aflags", 0), k.get("bflags", 0)
    names = {(256, 512, 0, ops.XF_PRELU | ops.XF_NORM): "heads_wgrad", (128, 512, 0, ops.XF_PRELU | ops.XF_NORM): "skip_head_wgrad",
             (512, 128, ops.XF_NORMBWD, 0): "bottleneck_wgrad", (1024, 128, 0, ops.XF_PRELU): "mask_wgrad",
             (128, 512, 0, ops.XF_NORM): "entry_bottleneck_wgrad"}
    label = names.get((M, N, af, bf), "nt_M{}_N{}_a{}_b{}".format(M, N, af, bf))
    rows = M + N + (M if k.get("A2") is not None else 0)
    return label, rows * T * 4.0 * batch


def instrumented_step(trainer, mixture, sources, full_step=True):
    """Run one step with HIP events around every GEMM / depthwise launch; returns per-family totals and, per launch
    shape, the launches with their durations in launch order.  full_step=False times the forward/backward pass only
    (no gradient exchange, no optimizer): what a multi-rank run uses, so that the instrumented pass contains no collective."""
    from speech_amd import ops
    rec = {"gemm_nn": [], "gemm_nt": [], "dw_fwd": [], "dw_bwd": []}
    orig = {k: getattr(ops, k) for k in rec}

    def wrap(name, work):
        fn = orig[name]

        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a, **k)
            e1.record()
            rec[name].append((e0, e1) + tuple(work(*a, **k)))
        return inner

    def dw_bwd_work(dn2, x, w, *a, **k):
        # three tensors (read x, read dy, write dx) when the launch recomputes the convolution output from x, four when
        # it reads the stored one (ops.dw_bwd's own rule decides, as in the launch)
        B, Cc, T = x.shape
        stat1, stat2 = k.get("stat1"), k.get("stat2")
        cln = (stat1 is not None and stat1.cln) or (stat2 is not None and stat2.cln)
        rc = bool(k.get("recompute_y2")) and T <= 4096 and w.shape[-1] <= 4 and not cln and stat1 is not None and stat2 is not None
        n = 3.0 if rc or k.get("y2") is None else 4.0
        kernel = "dw_bwd_gln_rc_kernel" if rc else ("dw_bwd_cln_kernel" if cln else "dw_bwd_gln_kernel")
        return n * B * Cc * T * 4, kernel, n * B * Cc * T * 4

    def dw_fwd_work(x, w, bias, y, *a, **k):
        B, Cc, T = x.shape
        acc = k.get("acc")
        cln = acc is not None and acc.col is not None
        kernel = "dw_fwd_cln_kernel" if cln else ("dw_fwd_row_kernel" if T <= 4096 and w.shape[-1] <= 4 else "dw_fwd_kernel")
        return 2.0 * B * Cc * T * 4, kernel, 2.0 * B * Cc * T * 4

    wrapped = {}
    wrapped["gemm_nn"] = wrap("gemm_nn", lambda M, T, K, batch, *a, **k: (2.0 * M * T * K * batch,) + _nn_label(M, T, K, batch, *a, **k))
    wrapped["gemm_nt"] = wrap("gemm_nt", lambda M, N, T, batch, *a, **k: (2.0 * M * N * T * batch,) + _nt_label(M, N, T, batch, *a, **k))
    wrapped["dw_fwd"] = wrap("dw_fwd", dw_fwd_work)
    wrapped["dw_bwd"] = wrap("dw_bwd", dw_bwd_work)
    run = trainer.step if full_step else trainer.forward_backward
    # per-kernel timing needs eager launches on ONE stream (sub-batch streams, when asked for, overlap kernels, which makes a
    # kernel's own duration meaningless); the work is the same full per-GPU batch
    graph_mode, trainer.use_graph = trainer.use_graph, False
    n_streams, trainer.n_streams = trainer.n_streams, 1
    try:
        # one eager step first, with the original entry points: the eager mode's activation buffers then come out of
        # torch's caching allocator as they do in every later step, and the timed launches do not pay for the first
        # touch of freshly mapped memory
        for k, f in orig.items():
            setattr(ops, k, f)
        run(mixture, sources)
        torch.cuda.synchronize()
        for k in rec:
            rec[k].clear()
        for k, f in wrapped.items():
            setattr(ops, k, f)
        run(mixture, sources)
        torch.cuda.synchronize()
    finally:
        trainer.use_graph, trainer.n_streams = graph_mode, n_streams
        for k, f in orig.items():
            setattr(ops, k, f)
    out = {}
    for name, lst in rec.items():
        launches = [(e0.elapsed_time(e1), work, label, nbytes) for e0, e1, work, label, nbytes in lst]
        out[name] = dict(launches=len(launches), ms=sum(l[0] for l in launches), work=sum(l[1] for l in launches),
                         sequence=launches)
    return out


def traffic_file(batch):
    """The committed rocprofv3 PMC summary of this benchmark at this per-GPU batch: the newest round's that exists."""
    for rnd in (3, 2):
        rel = "profiles/r{}_bench_b{}_pmc_traffic.json".format(rnd, batch)
        if os.path.exists(os.path.join(ROOT, rel)):
            return rel
    return "profiles/r3_bench_b{}_pmc_traffic.json".format(batch)


def _traffic_doc(batch):
    try:
        with open(os.path.join(ROOT, traffic_file(batch))) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    return d if d.get("batch_per_gpu") == batch else None


def profiled_step_bytes(batch):
    """Sum of the PMC-measured HBM bytes of every kernel of one pass (launches / passes x bytes per launch)."""
    d = _traffic_doc(batch)
    if d is None:
        return None
    passes = d.get("passes", 5)
    return sum(k["launches"] / passes * k["hbm_bytes_per_launch"] for k in d["kernels"].values())


def profiled_sequence(family, batch, n_launches):
    """PMC-measured HBM bytes of each launch of `family` in one pass, in launch order, or None when the summary has no
    such sequence or its length differs from this run's launch count (another schedule than the profiled one)."""
    d = _traffic_doc(batch)
    seq = None if d is None else d.get("sequence", {}).get(family)
    return seq if seq is not None and len(seq) == n_launches else None


def profiled_traffic(prefix, batch):
    """Average HBM bytes per launch of the kernels whose name contains `prefix`, from the committed rocprofv3 PMC
    summary of this benchmark at the same per-GPU batch (counters cannot be read inside the timed process);
    None when there is no summary for this batch."""
    d = _traffic_doc(batch)
    if d is None:
        return None
    n = tot = 0
    for name, k in d["kernels"].items():
        if prefix in name:
            n += k["launches"]
            tot += k["launches"] * k["hbm_bytes_per_launch"]
    return round(tot / n) if n else None


def _probe_rate(device, two_reads, mbytes=1024, reps=10):
    """GB/s of the library's own streaming probe (16-byte non-temporal loads and stores, nothing else) on buffers larger
    than the last-level cache: copy (one read, one write) or two reads and one write."""
    from speech_amd import _lib
    lib = _lib.load()
    n = mbytes * 1024 * 1024 // 4
    a = torch.empty(n, device=device, dtype=torch.float32).normal_()
    b = torch.empty_like(a).normal_() if two_reads else None
    y = torch.empty_like(a)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(lib.ctn_probe_copy(a.data_ptr(), b.data_ptr() if two_reads else None, y.data_ptr(), n, stream),
                   "ctn_probe_copy")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return (3.0 if two_reads else 2.0) * n * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def device_copy_rate(device):
    return _probe_rate(device, False)


def device_2r1w_rate(device):
    return _probe_rate(device, True)


def mfma_bare_loop_rate(device, iters=20000, bf16=False):
    """TFLOP/s of v_mfma_f32_32x32x2_f32 (bf16: v_mfma_f32_32x32x16_bf16) issued back to back from registers on random
    operands by every SIMD of the chip (four waves per SIMD, four independent accumulators per wave): the matrix rate
    this GPU sustains with no memory, LDS or vector work in the way."""
    from speech_amd import _lib
    lib = _lib.load()
    fn, name, flops = ((lib.ctn_probe_mfma_bf16, "ctn_probe_mfma_bf16", 32768) if bf16 else
                       (lib.ctn_probe_mfma_f32, "ctn_probe_mfma_f32", 4096))
    blocks = 256 * 4
    src = torch.empty(65536, device=device, dtype=torch.float32).normal_()
    out = torch.empty(blocks * 256, device=device, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(fn(src.data_ptr(), out.data_ptr(), blocks, 200, stream), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(fn(src.data_ptr(), out.data_ptr(), blocks, iters, stream), name)
    e1.record()
    torch.cuda.synchronize()
    return blocks * 4.0 * iters * 4 * flops / (e0.elapsed_time(e1) * 1e-3) / 1e12


def plain_gemm_rate(device, constant, batch=16, reps=30):
    """TFLOP/s of the plain 512 x 512 pointwise GEMM at the workload's frame count on random or constant operands: the
    f32 MFMA clock the chip sustains depends on how many operand bits toggle, so the rate on random data -- not the
    nominal peak -- is what a kernel of this design can reach on real activations."""
    from speech_amd import ops
    M = K = 512
    T = (T_SAMPLES - 16) // 8 + 1
    A = torch.empty(K, M, device=device)
    X = ops.new_act(batch, K, T, device)
    if constant:
        A.fill_(1.0); X.fill_(1.0)
    else:
        A.normal_(); X.normal_()
    Y = ops.new_act(batch, M, T, device)
    for _ in range(5):
        ops.gemm_nn(M, T, K, batch, A, M, X, Y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_nn(M, T, K, batch, A, M, X, Y)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * M * K * T * batch * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def cpu_baseline(cfg, steps=24, warmup=2, batch=2):
    """The oracle's training step (same arithmetic as the reference's loop) on the host CPU: a bounded sample of the
    bench workload, about ten seconds of CPU work at the reference recipe's per-GPU batch."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import convtasnet_oracle as O
    from speech_amd.utils import synth
    # threads: the share of the host this job may use (OMP_NUM_THREADS on a shared GPU box, else every core)
    cores = torch.get_num_threads()
    model_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model_name = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    sd = synth.fill_parameters(O.make_state(cfg), salt=1)
    opt = O.AdamState(sd)
    times = []
    for s in range(warmup + steps):
        mixture, sources = synth.make_batch(batch, cfg["n_sources"], T_SAMPLES, seed=2000 + s)
        t0 = time.perf_counter()
        O.train_step(mixture, sources, sd, cfg, opt)
        times.append(time.perf_counter() - t0)
    dt = sum(times[warmup:]) / steps
    return dict(value=round(batch / dt, 3), unit="utt/s", cores=cores, host_cpu_count=os.cpu_count(),
                cpu_model=model_name, kind="port",
                sample="{} training steps of batch {} (paper config, 4 s @ 8 kHz) after {} warm-up, "
                       "PyTorch CPU restatement of the reference step".format(steps, batch, warmup))


def shape_rows(prof, pieces, batch):
    """One row per launch shape of the instrumented step: launches, measured average duration, MFMA floor at the
    executed arithmetic's ceiling, HBM floor at 8 TB/s (PMC bytes where the committed summary has this run's launch
    sequence, algorithmic bytes otherwise)."""
    peak = executed_peak(pieces)
    rows = []
    for fam in ("gemm_nn", "gemm_nt"):
        seq = prof[fam]["sequence"]
        pmc = profiled_sequence(fam, batch, len(seq))
        groups = {}
        for i, (ms, flops, label, nbytes) in enumerate(seq):
            g = groups.setdefault(label, dict(n=0, ms=0.0, flops=0.0, alg=0.0, pmc=0.0))
            g["n"] += 1
            g["ms"] += ms
            g["flops"] += flops
            g["alg"] += nbytes
            g["pmc"] += pmc[i] if pmc is not None else 0.0
        for label, g in groups.items():
            us = g["ms"] * 1e3 / g["n"]
            flops, alg = g["flops"] / g["n"], g["alg"] / g["n"]
            hbm = g["pmc"] / g["n"] if pmc is not None else alg
            mfma_floor = flops / (peak * 1e12) * 1e6
            hbm_floor = hbm / (PEAK_HBM_GBS * 1e9) * 1e6
            rows.append(dict(shape=label, kernel=fam, launches_per_step=g["n"], avg_launch_us=round(us, 1),
                             gflop_per_launch=round(flops / 1e9, 2), algorithmic_mb=round(alg / 1e6, 1),
                             pmc_mb=round(hbm / 1e6, 1) if pmc is not None else None,
                             mfma_floor_us=round(mfma_floor, 1), hbm_floor_us=round(hbm_floor, 1),
                             bound="mfma" if mfma_floor >= hbm_floor else "hbm",
                             frac_of_floor=round(max(mfma_floor, hbm_floor) / us, 4)))
    rows.sort(key=lambda r: -r["launches_per_step"] * r["avg_launch_us"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="launch every kernel eagerly instead of replaying the forward/backward pass from a HIP graph")
    ap.add_argument("--streams", type=int, default=1,
                    help="sub-batches of a rank's shard run on separate HIP streams (1: the whole shard in one pass, "
                         "which is what the whole-output weight-gradient kernels are fastest at)")
    ap.add_argument("--buckets", type=int, default=4,
                    help="pieces of the flat gradient buffer all-reduced concurrently behind the pass (more than one rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-probe", action="store_true",
                    help="skip the second, informational measurement with split-bf16 GEMM products")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-eager", action="store_true",
                    help="more than one rank: also run the instrumented eager pass for the roofline rows (one-rank runs "
                         "always do); by default a multi-rank run measures scaling of the headline mode and nothing else")
    ap.add_argument("--no-recipe-batch", action="store_true",
                    help="skip the informational measurement at the reference recipe's batch of 2 per GPU")
    args = ap.parse_args()

    if os.environ.get("CTN_BENCH_TRACE_AFTER"):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CTN_BENCH_TRACE_AFTER"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # CTN_BENCH_REHEARSAL=gloo: every rank on cuda:0 with the gloo backend -- a rehearsal of the multi-rank control
    # flow (graph capture beside a process group, all-reduce, max-over-ranks timing) on a one-GPU box, not a result
    rehearsal = os.environ.get("CTN_BENCH_REHEARSAL") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.trainer import FlatTrainer
    from speech_amd.utils import synth

    cfg = PAPER
    model = build_model(cfg, device)
    # more than one rank: the gradient exchange follows the (graph-replayed) pass in `buckets` concurrent pieces; the
    # region-wise exchange overlapped with the backward pass stays off (FlatTrainer's default)
    trainer = FlatTrainer(model, PIT1d(NegSISDR(), cfg["n_sources"]), lr=1e-3, max_norm=5.0, n_streams=args.streams,
                          use_graph=args.graph, n_buckets=args.buckets if world > 1 else 1)
    n_batches = 4
    batches = []
    for i in range(n_batches):
        mixture, sources = synth.make_batch(args.batch, cfg["n_sources"], T_SAMPLES, seed=1000 * rank + i)
        batches.append((mixture.to(device), sources.to(device)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, outside the W warm-up and K timed steps: library load, lazy caches and (graph mode) the capture,
    # which FlatTrainer performs on its third step for a given input shape
    for i in range(3):
        trainer.step(*batches[i % n_batches])
    losses = []
    for i in range(args.warmup):
        loss, _ = trainer.step(*batches[i % n_batches])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = trainer.step(*batches[i % n_batches])
        losses.append(loss)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device="cpu" if rehearsal else device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    final_loss = float(losses[-1].item()) if losses else float("nan")

    from speech_amd import ops as _ops
    headline_split = _ops.get_gemm_split()
    utts = args.batch * world * args.steps
    step_ms = dt / args.steps * 1e3
    # ---- the headline: complete from here on.  Everything below adds informational fields; each part is guarded so
    # that a failure there cannot cost the measured value, and none of it contains a collective.
    line = {
        "metric": "Conv-TasNet train utt/s (4 s, 8 kHz, 2-spk)",
        "value": round(utts / dt, 3),
        "unit": "utt/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(step_ms, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "arithmetic": {0: "f32 MFMA (v_mfma_f32_32x32x2_f32)",
                       2: "fp32 operands split into two bf16 pieces, three products on the bf16 MFMA, fp32 accumulation",
                       3: "fp32 in, fp32 out, fp32 accumulation; GEMM products from an EXACT split of every fp32 operand "
                          "into three bf16 pieces, six products on v_mfma_f32_32x32x16_bf16 (what is dropped is below one "
                          "fp32 rounding of the product); every parity test of the suite runs in this mode"}[headline_split],
        "data": "synthetic",
        "config": {"workload": "Conv-TasNet paper config N512 L16 B128 H512 Sc128 P3 X8 R3 gLN, 2 speakers, "
                               "32000-sample (4 s @ 8 kHz) mixtures, full training step "
                               "(fwd + PIT SI-SDR + bwd + all-reduce + clip 5 + Adam 1e-3)",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "parallelism": "dp{}".format(world), "sub_batch_streams": args.streams,
                   "hip_graph": bool(trainer.use_graph), "final_loss": round(final_loss, 4),
                   "exchange": ("none (one rank)" if world == 1 else
                                "after the {} pass, {} bucket(s) of the flat fp32 gradient buffer in flight together"
                                .format("replayed" if trainer.use_graph else "eager", trainer.n_buckets))},
    }
    errors = {}

    def guarded(name, fn):
        try:
            return fn()
        except Exception as exc:                       # noqa: BLE001 -- informational parts must not cost the headline
            errors[name] = "{}: {}".format(type(exc).__name__, exc)
            return None

    one_rank = world == 1

    # Informational extra measurements, never `value`: the same K steps in the two other arithmetic modes of the GEMMs
    # (CTN_ARITH_* in include/ctn.h) -- plain f32 MFMAs, and two bf16 pieces per operand.  One-rank runs only.
    def other_modes():
        notes = {0: "informational, not the headline: the same step with the GEMM products on v_mfma_f32_32x32x2_f32",
                 2: "opt-in throughput mode, not the headline: GEMM products as hi*hi + hi*lo + lo*hi of two bf16 pieces "
                    "per operand on the bf16 MFMA, fp32 accumulation; each GEMM within 5e-5 instead of 2e-5 of exact "
                    "math, waveforms and loss within 1e-4 of the reference (tests/test_split_gpu.py)",
                 3: "informational: three bf16 pieces per operand (an exact representation of the fp32 value) and the "
                    "six leading products on the bf16 MFMA, fp32 accumulation"}
        names = {0: "f32_mfma", 2: "bf16_two_pieces", 3: "bf16_three_pieces_exact"}
        probe = {}
        try:
            for pieces in (0, 2, 3):
                if pieces == headline_split:
                    continue
                _ops.set_gemm_split(pieces)
                trainer._graphs.clear()            # captured graphs hold the kernels of the mode they were captured in
                trainer._eager_steps = 0
                for i in range(3 + args.warmup):
                    trainer.step(*batches[i % n_batches])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    loss_s, _ = trainer.step(*batches[i % n_batches])
                torch.cuda.synchronize()
                dts = time.perf_counter() - t1
                probe[names[pieces]] = dict(
                    value=round(args.batch * args.steps / dts, 3), unit="utt/s",
                    ms_per_step=round(dts / args.steps * 1e3, 3), final_loss=round(float(loss_s.item()), 4),
                    note=notes[pieces])
        finally:
            _ops.set_gemm_split(headline_split)
            trainer._graphs.clear()
            trainer._eager_steps = 0
        return probe

    # The reference recipe's own per-GPU batch (egs/wsj0-mix/conv-tasnet/train.sh:54: batch_size=2), same step, same
    # mode: informational, never `value`.
    def recipe_batch():
        small = []
        for i in range(n_batches):
            mixture, sources = synth.make_batch(2, cfg["n_sources"], T_SAMPLES, seed=5000 + i)
            small.append((mixture.to(device), sources.to(device)))
        keep_streams, trainer.n_streams = trainer.n_streams, 1      # two utterances are one sub-batch
        n_small = max(args.steps, 50)
        try:
            for i in range(3 + args.warmup):
                trainer.step(*small[i % n_batches])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(n_small):
                trainer.step(*small[i % n_batches])
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
        finally:
            trainer.n_streams = keep_streams
        return dict(value=round(2 * n_small / dt2, 3), unit="utt/s", per_gpu_batch=2, steps=n_small,
                    ms_per_step=round(dt2 / n_small * 1e3, 3),
                    note="the reference recipe's batch size per GPU; a step is ~300 launches that no longer fill the chip")

    line["recipe_batch_2"] = (guarded("recipe_batch_2", recipe_batch)
                              if one_rank and not args.no_recipe_batch and args.batch != 2 else None)
    line["other_arithmetic_modes"] = (guarded("other_arithmetic_modes", other_modes)
                                      if one_rank and not args.no_split_probe else None)

    # The roofline rows come from an instrumented EAGER pass.  One rank: a whole step.  More ranks: only on request
    # (--profile-eager), and then the forward/backward pass alone, which contains no collective -- every rank may fail or
    # succeed on its own without leaving the others waiting.
    want_roofline = not args.no_roofline and (one_rank or args.profile_eager)
    prof = guarded("instrumented_step", lambda: instrumented_step(trainer, *batches[0], full_step=one_rank)) \
        if want_roofline else None
    prof_f32 = None
    if prof is not None and headline_split != 0 and rank == 0:
        # the same instrumented pass with the products on f32 MFMAs: the instruction the f32 matrix peak is quoted for
        def f32_pass():
            _ops.set_gemm_split(0)
            try:
                return instrumented_step(trainer, *batches[0], full_step=one_rank)
            finally:
                _ops.set_gemm_split(headline_split)
        prof_f32 = guarded("instrumented_step_f32", f32_pass) if one_rank else None

    def roofline_rows():
        def gemm_rows(pr, pieces):
            peak = executed_peak(pieces)
            rows = {}
            for key, kernel in (("gemm_nn", "gemm_nn_fast_kernel (pointwise conv fwd + dgrad)"),
                                ("gemm_nt", "gemm_nt_whole_kernel + its reduction, gemm_nt_fast_kernel (weight gradients)")):
                g = pr[key]
                rate = g["work"] / (g["ms"] * 1e-3) / 1e12
                row = dict(bound="mfma", kernel=kernel, achieved=round(rate, 2), peak=round(peak, 1), unit="TFLOP/s",
                           frac=round(rate / peak, 4), launches_per_step=g["launches"],
                           avg_launch_us=round(g["ms"] * 1e3 / g["launches"], 1),
                           frac_of_f32_matrix_peak=round(rate / PEAK_F32_MFMA_TFLOPS, 4))
                if pieces:
                    n = PRODUCTS[pieces]
                    row["instructions"] = "v_mfma_f32_32x32x16_bf16, {} per algorithmic fp32 product tile".format(n)
                    row["peak_is"] = ("{} TFLOP/s dense bf16 matrix peak / {} matrix instructions per fp32 product"
                                      .format(PEAK_BF16_MFMA_TFLOPS, n))
                    row["bf16_mfma_issue_rate"] = dict(achieved=round(rate * n, 1), peak=PEAK_BF16_MFMA_TFLOPS,
                                                       frac=round(rate * n / PEAK_BF16_MFMA_TFLOPS, 4), unit="TFLOP/s")
                else:
                    row["instructions"] = "v_mfma_f32_32x32x2_f32"
                    row["peak_is"] = "the dense f32 matrix peak"
                rows[key] = row
            return rows

        out = {}
        rows = gemm_rows(prof, headline_split)
        roof = rows["gemm_nn"]
        bare32 = guarded("mfma_bare_loop", lambda: round(mfma_bare_loop_rate(device), 1))
        bare16 = guarded("bf16_mfma_bare_loop", lambda: round(mfma_bare_loop_rate(device, bf16=True), 1))
        roof.update(
            note="achieved = algorithmic fp32 FLOPs (2*M*K*T' per launch and utterance) / summed launch durations; peak = the "
                 "ceiling of the arithmetic that executes (peak_is), so frac <= 1 by construction",
            traffic=profiled_traffic("gemm_nn_fast_kernel", args.batch), traffic_source=traffic_file(args.batch),
            mode="one instrumented step of the same batch, single stream, eager launches, after one eager warm-up step "
                 "(bench.py --streams 1 --no-graph is the profiled command)",
            mfma_bare_loop=bare32, bf16_mfma_bare_loop=bare16)
        if bare16 and headline_split:
            roof["frac_of_measured_bare_loop"] = round(roof["achieved"] * PRODUCTS[headline_split] / bare16, 4)
        out["roofline"] = roof
        out["roofline_wgrad"] = rows["gemm_nt"]
        out["roofline_shapes"] = shape_rows(prof, headline_split, args.batch)
        if prof_f32 is not None:
            rows32 = gemm_rows(prof_f32, 0)
            _ops.set_gemm_split(0)
            try:
                rows32["gemm_nn"]["plain_gemm_on_random_operands"] = guarded(
                    "plain_gemm_random", lambda: round(plain_gemm_rate(device, False), 1))
                rows32["gemm_nn"]["plain_gemm_on_constant_operands"] = guarded(
                    "plain_gemm_constant", lambda: round(plain_gemm_rate(device, True), 1))
            finally:
                _ops.set_gemm_split(headline_split)
            out["roofline_f32_mfma_mode"] = dict(pointwise=rows32["gemm_nn"], wgrad=rows32["gemm_nt"],
                                                 kernel_ms_per_step={k: round(v["ms"], 3) for k, v in prof_f32.items()})
        copy_gbs = guarded("copy_probe", lambda: device_copy_rate(device))
        rw2_gbs = guarded("two_read_one_write_probe", lambda: device_2r1w_rate(device))
        hb = []
        for name in ("dw_fwd", "dw_bwd"):
            d = prof[name]
            if not d["launches"]:
                continue
            gbs = d["work"] / (d["ms"] * 1e-3) / 1e9
            kernels = sorted(set(l[2] for l in d["sequence"]))
            per_tensor = d["work"] / d["launches"] / (args.batch * cfg["sep_hidden_channels"] * ((T_SAMPLES - 16) // 8 + 1) * 4.0)
            row = dict(bound="hbm", kernel=" / ".join(kernels), achieved=round(gbs, 1), peak=PEAK_HBM_GBS,
                       unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                       traffic=profiled_traffic(name, args.batch),
                       launches_per_step=d["launches"], avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 1),
                       bytes_priced="{:.0f} x H x T' x 4 per utterance ({})".format(
                           per_tensor, "read y1, write y2" if name == "dw_fwd" else
                           ("read x, read dy, write dx; the convolution output is recomputed from x" if per_tensor < 3.5
                            else "read x, read the stored convolution output, read dy, write dx")),
                       copy_kernel_rate=round(copy_gbs, 1) if copy_gbs else None,
                       two_read_one_write_kernel_rate=round(rw2_gbs, 1) if rw2_gbs else None,
                       frac_of_copy_kernel=round(gbs / copy_gbs, 4) if copy_gbs else None)
            hb.append(row)
        out["roofline_hbm"] = hb
        out["kernel_ms_per_step"] = {k: round(v["ms"], 3) for k, v in prof.items()}
        out["kernel_ms_per_step"]["step"] = round(step_ms, 3)
        # the whole step against both roofs
        flops_step = prof["gemm_nn"]["work"] + prof["gemm_nt"]["work"]
        pmc_step = profiled_step_bytes(args.batch)
        alg_step = sum(l[3] for fam in prof.values() for l in fam["sequence"])
        rs = dict(ms_per_step=round(step_ms, 3),
                  gemm_gflop_per_step=round(flops_step / 1e9, 1),
                  mfma_floor_ms=round(flops_step / (executed_peak(headline_split) * 1e12) * 1e3, 2),
                  algorithmic_gb_per_step_gemm_and_depthwise=round(alg_step / 1e9, 1),
                  pmc_gb_per_step=round(pmc_step / 1e9, 1) if pmc_step else None,
                  pmc_source=traffic_file(args.batch))
        if pmc_step:
            rs["achieved_tb_per_s"] = round(pmc_step / (step_ms * 1e-3) / 1e12, 2)
            rs["frac_of_hbm_peak"] = round(pmc_step / (step_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4)
            rs["hbm_floor_ms"] = round(pmc_step / (PEAK_HBM_GBS * 1e9) * 1e3, 2)
            rs["bound"] = "hbm" if rs["hbm_floor_ms"] >= rs["mfma_floor_ms"] else "mfma"
            rs["frac_of_floor"] = round(max(rs["hbm_floor_ms"], rs["mfma_floor_ms"]) / step_ms, 4)
        out["roofline_step"] = rs
        return out

    roof_fields = guarded("roofline", roofline_rows) if (rank == 0 and prof is not None) else None
    if roof_fields is None:
        roof_fields = {"roofline": None, "roofline_hbm": None}
        if world > 1 and not args.profile_eager:
            roof_fields["roofline_note"] = ("multi-rank runs measure scaling only; the roofline rows are in the one-GPU "
                                            "line (or add --profile-eager)")
    line.update(roof_fields)
    if rank == 0:
        line["cpu_baseline"] = guarded("cpu_baseline", lambda: cpu_baseline(cfg)) if one_rank and not args.no_cpu_baseline else None
        if errors:
            line["errors"] = errors
        print(json.dumps(line), flush=True)
    if world > 1:
        guarded("barrier", dist.barrier)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
