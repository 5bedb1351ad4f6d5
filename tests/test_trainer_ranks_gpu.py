"""FlatTrainer with two ranks: both on cuda:0 over gloo (the multi-rank control flow of the data-parallel step -- weight
broadcast, region-wise exchange during the backward pass, bucketed exchange after a graph replay, clip after the
reduction -- without a second GPU).  Every exchange schedule must leave both ranks with the same parameters, and those
must equal a single process's update on the global batch (egs/wsj0-mix/common/src/driver.py:146-155 semantics)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, n_streams, use_graph, overlap_exchange, n_buckets)
VARIANTS = [("eager_overlap", 1, False, True, 1), ("eager_after", 1, False, False, 1),
            ("streams2_overlap", 2, False, True, 1), ("graph_buckets4", 2, True, None, 4)]
STEPS = 4


def _paths():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _train(variant, world, rank, device):
    import cases
    import convtasnet_oracle as O
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.trainer import FlatTrainer, shard_batch
    from test_parity_gpu import build, load_state
    _, n_streams, use_graph, overlap, n_buckets = variant
    cfg = dict(O.CONFIG_SMALL, sep_num_blocks=3, sep_num_layers=2)       # three regions
    model = load_state(build(cfg), cases.e2e_state(cfg))
    if rank > 0:                                   # the constructor's broadcast must make the replicas identical
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
    tr = FlatTrainer(model, PIT1d(NegSISDR(), 2), lr=1e-3, max_norm=5.0, n_streams=n_streams, use_graph=use_graph,
                     overlap_exchange=overlap, n_buckets=n_buckets)
    losses = []
    for i in range(STEPS):
        mixture, sources = cases.e2e_batch(cfg, 4, 4000, 50 + i)
        if world > 1:
            mixture, sources = shard_batch(mixture, sources, rank, world)
        loss, _ = tr.step(mixture.to(device), sources.to(device))
        losses.append(loss.item())
    torch.cuda.synchronize()
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, losses


def _epoch_losses(world, rank, device, out_dir):
    """One epoch of the epoch-loop trainer (speech_amd.driver.AdhocTrainer): training loss and validation loss.  The
    validation set has 5 utterances; with two ranks, rank 0 sees 3 of them and rank 1 sees 2 (uneven shards, ragged last
    batch), and the loss both ranks report must be the mean over all 5."""
    import types
    import cases
    import convtasnet_oracle as O
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.driver import AdhocTrainer
    from speech_amd.trainer import shard_batch
    from test_parity_gpu import build, load_state
    cfg = dict(O.CONFIG_SMALL, sep_num_blocks=1, sep_num_layers=2)
    model = load_state(build(cfg), cases.e2e_state(cfg)).to(device)
    train = []
    for i in range(2):
        mixture, sources = cases.e2e_batch(cfg, 4, 4000, 70 + i)
        if world > 1:
            mixture, sources = shard_batch(mixture, sources, rank, world)
        train.append((mixture, sources))
    vm, vs = cases.e2e_batch(cfg, 5, 4000, 90)
    mine = list(range(5)) if world == 1 else ([0, 1, 2] if rank == 0 else [3, 4])
    valid = [(vm[mine[j:j + 2]], vs[mine[j:j + 2]]) for j in range(0, len(mine), 2)]     # batches of two, the last ragged
    tag = "w{}r{}".format(world, rank)
    args = types.SimpleNamespace(sample_rate=8000, n_sources=2, max_norm=5.0, epochs=1, use_cuda=True, continue_from=None,
                                 overwrite=True, model_dir=os.path.join(out_dir, tag, "model"),
                                 loss_dir=os.path.join(out_dir, tag, "loss"), sample_dir=os.path.join(out_dir, tag, "sample"))
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    tr = AdhocTrainer(model, {"train": train, "valid": valid}, PIT1d(NegSISDR(), 2), optimizer, args)
    return [float(v) for v in tr.run_one_epoch(0)]


def _worker(rank, world, port, out_dir):
    _paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda", 0)
    out = {}
    for variant in VARIANTS:
        params, losses = _train(variant, world, rank, device)
        out[variant[0]] = dict(params=params, losses=losses)
        dist.barrier()
    out["epoch"] = _epoch_losses(world, rank, device, out_dir)
    dist.barrier()
    torch.save(out, os.path.join(out_dir, "rank{}.pt".format(rank)))
    dist.destroy_process_group()


def test_two_ranks_every_exchange_schedule_equals_global_batch(tmp_path):
    _paths()
    world, port = 2, 29700 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    ranks = [torch.load(os.path.join(str(tmp_path), "rank{}.pt".format(r)), weights_only=True) for r in range(world)]
    single, single_losses = _train(("single", 1, False, False, 1), 1, 0, torch.device("cuda", 0))
    # the epoch loop: both ranks report the same losses, equal to one process on the whole training and validation sets
    single_epoch = _epoch_losses(1, 0, torch.device("cuda", 0), str(tmp_path))
    assert ranks[0]["epoch"] == ranks[1]["epoch"]
    for a, b in zip(ranks[0]["epoch"], single_epoch):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (ranks[0]["epoch"], single_epoch)
    for name, *_ in VARIANTS:
        p0, p1 = ranks[0][name]["params"], ranks[1][name]["params"]
        mean_loss = [(a + b) / 2 for a, b in zip(ranks[0][name]["losses"], ranks[1][name]["losses"])]
        for a, b in zip(mean_loss, single_losses):     # mean of the shard means = global-batch mean
            assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (name, mean_loss, single_losses)
        for k in single:
            assert torch.equal(p0[k], p1[k]), "{}: ranks diverged on {}".format(name, k)
            d = (p0[k] - single[k]).abs().max().item()
            # four Adam steps; gradients are summed in a different order (atomics, two shards), and a step moves a weight by
            # about lr * sign-like(g), so a rounding-level change of a near-zero gradient component shows at the 1e-5 level
            assert d <= 2e-5, "{}: {} differs from the global-batch update by {}".format(name, k, d)


def _bench_worker(rank, world, port, out_dir, extra):
    """bench.main() as one rank of a two-rank job: every rank on cuda:0 over gloo (CTN_BENCH_REHEARSAL), the command
    line the scaling run uses, rank 0's stdout captured to a file."""
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), CTN_BENCH_REHEARSAL="gloo")
    import contextlib
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "2"] + list(extra)
    with open(os.path.join(out_dir, "rank{}.out".format(rank)), "w") as f, contextlib.redirect_stdout(f):
        bench.main()


@pytest.mark.parametrize("extra", [(), ("--profile-eager",)], ids=["default", "profile_eager"])
def test_bench_two_rank_rehearsal_prints_one_complete_line(tmp_path, extra):
    """The multi-rank control flow of bench.py (process group, weight broadcast, graph capture beside the group, bucketed
    exchange behind the replay, max-over-ranks timing, the guarded informational part) with two ranks on the one GPU:
    rank 0 prints exactly one JSON line with the contract's fields, the other rank prints nothing."""
    import json
    world, port = 2, 31700 + (os.getpid() % 2000)
    mp.start_processes(_bench_worker, args=(world, port, str(tmp_path), extra), nprocs=world, join=True, start_method="spawn")
    lines = [ln for ln in open(os.path.join(str(tmp_path), "rank0.out")).read().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert open(os.path.join(str(tmp_path), "rank1.out")).read().strip() == ""
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 2 * 2 * 2 / (line["ms_per_step"] * 2e-3)) <= 1e-2 * line["value"]
    assert line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2"
    assert "4 bucket" in line["config"]["exchange"] and line["config"]["hip_graph"] is True
    assert "errors" not in line, line.get("errors")
    if extra:
        assert line["roofline"]["frac"] <= 1.0 and line["roofline_shapes"]
    else:
        assert line["roofline"] is None
