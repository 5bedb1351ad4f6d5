"""Benchmark: Conv-TasNet training utterances/s (paper config, 2 speakers, 4 s @ 8 kHz) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward -> PIT SI-SDR -> backward -> gradient all-reduce (N > 1) -> clip(5) -> Adam(1e-3) on one
synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line.  Weak scaling: the per-GPU batch is
fixed, value = utterances all ranks processed / max-over-ranks time.

The line also carries
  roofline      live HIP-event timing of the dominant kernel (the pointwise-conv GEMM, gemm_nn_fast_kernel) over one
                instrumented step: algorithmic fp32 FLOPs of its launches / their summed duration, vs the 157.3
                TFLOP/s dense f32 matrix peak of MI355X (the path computes in fp32; in the default arithmetic each
                product is formed exactly from six bf16 matrix instructions, whose own issue rate the row also carries);
  roofline_hbm  the same for the fused depthwise kernels against the 8 TB/s HBM3E peak
                (algorithmic bytes: forward 2 x H x T' x 4, backward 3 x H x T' x 4 per utterance and layer), with the
                rate of a hand-written 16-byte
                copy kernel and of the bare f32-MFMA loop measured in the same process as yardsticks;
  cpu_baseline  the CPU oracle (a PyTorch restatement of the reference's step) timed on this host.
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
PEAK_HBM_GBS = 8000.0


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


def instrumented_step(trainer, mixture, sources):
    """Run one step with HIP events around every GEMM / depthwise launch; returns per-kernel totals."""
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
            rec[name].append((e0, e1, work(*a, **k)))
        return inner

    wrapped = {}
    wrapped["gemm_nn"] = wrap("gemm_nn", lambda M, T, K, batch, *a, **k: 2.0 * M * T * K * batch)
    wrapped["gemm_nt"] = wrap("gemm_nt", lambda M, N, T, batch, *a, **k: 2.0 * M * N * T * batch)
    wrapped["dw_fwd"] = wrap("dw_fwd", lambda x, w, bias, y, *a, **k: 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * 4)
    # the backward kernel is priced at 3 tensors (read x, read dy, write dx), which is what it moves: the convolution
    # output is recomputed from x instead of being read
    wrapped["dw_bwd"] = wrap("dw_bwd", lambda dn2, x, *a, **k: 3.0 * x.shape[0] * x.shape[1] * x.shape[2] * 4)
    # per-kernel timing needs eager launches on ONE stream (two sub-batch streams overlap kernels, which makes a
    # kernel's own duration meaningless); the work is the same full per-GPU batch
    graph_mode, trainer.use_graph = trainer.use_graph, False
    n_streams, trainer.n_streams = trainer.n_streams, 1
    try:
        # one eager step first, with the original entry points: the eager mode's activation buffers then come out of
        # torch's caching allocator as they do in every later step, and the timed launches do not pay for the first
        # touch of freshly mapped memory
        for k, f in orig.items():
            setattr(ops, k, f)
        trainer.step(mixture, sources)
        torch.cuda.synchronize()
        for k in rec:
            rec[k].clear()
        ops.gemm_nn, ops.gemm_nt, ops.dw_fwd, ops.dw_bwd = wrapped["gemm_nn"], wrapped["gemm_nt"], wrapped["dw_fwd"], wrapped["dw_bwd"]
        trainer.step(mixture, sources)
        torch.cuda.synchronize()
    finally:
        trainer.use_graph, trainer.n_streams = graph_mode, n_streams
        for k, f in orig.items():
            setattr(ops, k, f)
    out = {}
    for name, lst in rec.items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in lst)
        out[name] = dict(launches=len(lst), ms=ms, work=sum(w for _, _, w in lst))
    return out


def traffic_file(batch):
    return "profiles/r2_bench_b{}_pmc_traffic.json".format(batch)


def profiled_traffic(prefix, batch):
    """Average HBM bytes per launch of the kernels whose name contains `prefix`, from the committed rocprofv3 PMC
    summary of this benchmark at the same per-GPU batch (counters cannot be read inside the timed process);
    None when there is no summary for this batch."""
    path = os.path.join(ROOT, traffic_file(batch))
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("batch_per_gpu") != batch:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="launch every kernel eagerly instead of replaying the forward/backward pass from a HIP graph")
    ap.add_argument("--streams", type=int, default=2, help="sub-batches of a rank's shard run on separate HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-probe", action="store_true",
                    help="skip the second, informational measurement with split-bf16 GEMM products")
    ap.add_argument("--no-roofline", action="store_true")
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
    trainer = FlatTrainer(model, PIT1d(NegSISDR(), cfg["n_sources"]), lr=1e-3, max_norm=5.0, n_streams=args.streams, use_graph=args.graph)
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

    # Informational extra measurements, never `value`: the same K steps in the two other arithmetic modes of the GEMMs
    # (CTN_ARITH_* in include/ctn.h) -- plain f32 MFMAs, and two bf16 pieces per operand.  One-GPU runs only: the
    # multi-GPU runs measure scaling of the headline mode and nothing else.
    from speech_amd import ops as _ops
    headline_split = _ops.get_gemm_split()
    split_probe = None
    if world == 1 and not args.no_split_probe:
        notes = {0: "informational, not the headline: the same step with the GEMM products on v_mfma_f32_32x32x2_f32",
                 2: "opt-in throughput mode, not the headline: GEMM products as hi*hi + hi*lo + lo*hi of two bf16 pieces "
                    "per operand on the bf16 MFMA, fp32 accumulation; each GEMM within 5e-5 instead of 2e-5 of exact "
                    "math, waveforms and loss within 1e-4 of the reference (tests/test_split_gpu.py)",
                 3: "informational: three bf16 pieces per operand (an exact representation of the fp32 value) and the "
                    "six leading products on the bf16 MFMA, fp32 accumulation"}
        names = {0: "f32_mfma", 2: "bf16_two_pieces", 3: "bf16_three_pieces_exact"}
        split_probe = {}
        try:
            for pieces in (0, 2, 3):
                if pieces == headline_split:
                    continue
                _ops.set_gemm_split(pieces)
                trainer._graphs.clear()            # captured graphs hold the kernels of the mode they were captured in
                trainer._eager_steps = 0
                for i in range(3 + args.warmup):
                    trainer.step(*batches[i % n_batches])
                barrier()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    loss_s, _ = trainer.step(*batches[i % n_batches])
                barrier()
                dts = time.perf_counter() - t1
                split_probe[names[pieces]] = dict(
                    value=round(args.batch * world * args.steps / dts, 3), unit="utt/s",
                    ms_per_step=round(dts / args.steps * 1e3, 3), final_loss=round(float(loss_s.item()), 4),
                    note=notes[pieces])
        finally:
            _ops.set_gemm_split(headline_split)
            trainer._graphs.clear()
            trainer._eager_steps = 0

    # The reference recipe's own per-GPU batch (egs/wsj0-mix/conv-tasnet/train.sh:54: batch_size=2), same step, same
    # mode: informational, never `value`.
    recipe = None
    if world == 1 and not args.no_recipe_batch and args.batch != 2:
        small = []
        for i in range(n_batches):
            mixture, sources = synth.make_batch(2, cfg["n_sources"], T_SAMPLES, seed=5000 + i)
            small.append((mixture.to(device), sources.to(device)))
        keep_streams, trainer.n_streams = trainer.n_streams, 1      # two utterances are one sub-batch
        n_small = max(args.steps, 50)
        try:
            for i in range(3 + args.warmup):
                trainer.step(*small[i % n_batches])
            barrier()
            t2 = time.perf_counter()
            for i in range(n_small):
                trainer.step(*small[i % n_batches])
            barrier()
            dt2 = time.perf_counter() - t2
        finally:
            trainer.n_streams = keep_streams
        recipe = dict(value=round(2 * n_small / dt2, 3), unit="utt/s", per_gpu_batch=2, steps=n_small,
                      ms_per_step=round(dt2 / n_small * 1e3, 3),
                      note="the reference recipe's batch size per GPU; a step is ~300 launches that no longer fill the chip")

    roof = roof_hbm = None
    extra = {}
    prof = None
    if not args.no_roofline:
        # every rank runs the instrumented step: it contains the gradient all-reduce, a collective
        prof = instrumented_step(trainer, *batches[0])
    prof_f32 = None
    if prof is not None and headline_split != 0:
        # the same instrumented step with the products on f32 MFMAs: the instruction the f32 matrix peak is quoted for
        _ops.set_gemm_split(0)
        try:
            prof_f32 = instrumented_step(trainer, *batches[0])
        finally:
            _ops.set_gemm_split(headline_split)
    if rank == 0 and prof is not None:
        PEAK_BF16_MFMA_TFLOPS = 2500.0
        products = {0: 1, 2: 3, 3: 6}[headline_split]      # bf16 instruction work per algorithmic product

        def gemm_rows(pr, pieces):
            rows = {}
            for key, kernel in (("gemm_nn", "gemm_nn_fast_kernel (pointwise conv fwd + dgrad)"),
                                ("gemm_nt", "gemm_nt_fast_kernel (weight gradients)")):
                g = pr[key]
                rate = g["work"] / (g["ms"] * 1e-3) / 1e12
                row = dict(bound="mfma", kernel=kernel, achieved=round(rate, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                           frac=round(rate / PEAK_F32_MFMA_TFLOPS, 4), launches_per_step=g["launches"],
                           avg_launch_us=round(g["ms"] * 1e3 / g["launches"], 1))
                if pieces:
                    n = {2: 3, 3: 6}[pieces]
                    row["instructions"] = "v_mfma_f32_32x32x16_bf16, {} per algorithmic fp32 product tile".format(n)
                    row["bf16_mfma_issue_rate"] = dict(achieved=round(rate * n, 1), peak=PEAK_BF16_MFMA_TFLOPS,
                                                       frac=round(rate * n / PEAK_BF16_MFMA_TFLOPS, 4), unit="TFLOP/s")
                else:
                    row["instructions"] = "v_mfma_f32_32x32x2_f32"
                rows[key] = row
            return rows

        rows = gemm_rows(prof, headline_split)
        roof = rows["gemm_nn"]
        roof.update(
            note="achieved = algorithmic fp32 FLOPs (2*M*K*T' per launch and utterance) / summed launch durations; peak = the "
                 "dense f32 matrix peak, the peak of the dtype the path computes in",
            traffic=profiled_traffic("gemm_nn_fast_kernel", args.batch), traffic_source=traffic_file(args.batch),
            mode="one instrumented step of the same batch, single stream, eager launches, after one eager warm-up step "
                 "(bench.py --streams 1 --no-graph is the profiled command)",
            mfma_bare_loop=round(mfma_bare_loop_rate(device), 1),
            bf16_mfma_bare_loop=round(mfma_bare_loop_rate(device, bf16=True), 1))
        extra["roofline_wgrad"] = rows["gemm_nt"]
        if prof_f32 is not None:
            rows32 = gemm_rows(prof_f32, 0)
            rows32["gemm_nn"].update(plain_gemm_on_random_operands=None, plain_gemm_on_constant_operands=None)
            _ops.set_gemm_split(0)
            try:
                rows32["gemm_nn"]["plain_gemm_on_random_operands"] = round(plain_gemm_rate(device, False), 1)
                rows32["gemm_nn"]["plain_gemm_on_constant_operands"] = round(plain_gemm_rate(device, True), 1)
            finally:
                _ops.set_gemm_split(headline_split)
            extra["roofline_f32_mfma_mode"] = dict(pointwise=rows32["gemm_nn"], wgrad=rows32["gemm_nt"],
                                                   kernel_ms_per_step={k: round(v["ms"], 3) for k, v in prof_f32.items()})
        copy_gbs = device_copy_rate(device)
        rw2_gbs = device_2r1w_rate(device)
        hb = []
        for name in ("dw_fwd", "dw_bwd"):
            d = prof[name]
            gbs = d["work"] / (d["ms"] * 1e-3) / 1e9
            row = dict(bound="hbm", kernel={"dw_fwd": "dw_fwd_row_kernel", "dw_bwd": "dw_bwd_gln_rc_kernel"}[name], achieved=round(gbs, 1), peak=PEAK_HBM_GBS,
                       unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                       traffic=profiled_traffic(name, args.batch),
                       launches_per_step=d["launches"], avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 1),
                       bytes_priced="2 x H x T' x 4 per utterance (read y1, write y2)" if name == "dw_fwd" else
                                    "3 x H x T' x 4 per utterance (read x, read dy, write dx)",
                       copy_kernel_rate=round(copy_gbs, 1), two_read_one_write_kernel_rate=round(rw2_gbs, 1),
                       frac_of_copy_kernel=round(gbs / copy_gbs, 4))
            if name == "dw_bwd":
                row["note"] = ("the kernel recomputes the convolution output from x instead of reading the stored one, so "
                               "it moves the three tensors it is priced at")
            hb.append(row)
        roof_hbm = hb
        step_ms = dt / args.steps * 1e3
        extra["kernel_ms_per_step"] = {k: round(v["ms"], 3) for k, v in prof.items()}
        extra["kernel_ms_per_step"]["step"] = round(step_ms, 3)
    if world > 1:
        dist.barrier()

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg)
        utts = args.batch * world * args.steps
        line = {
            "metric": "Conv-TasNet train utt/s (4 s, 8 kHz, 2-spk)",
            "value": round(utts / dt, 3),
            "unit": "utt/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
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
                       "hip_graph": bool(args.graph), "final_loss": round(final_loss, 4)},
            "recipe_batch_2": recipe,
            "other_arithmetic_modes": split_probe,
            "roofline": roof,
            "roofline_hbm": roof_hbm,
            "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
