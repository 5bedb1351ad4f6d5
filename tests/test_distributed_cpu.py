"""Two gloo ranks on the CPU: the data-parallel recipe of speech_amd.trainer (equal batch shards, sum
all-reduce of the flat gradient, scale by 1/world, clip AFTER the reduction) reproduces the single-process
global-batch update of the reference loop (egs/wsj0-mix/common/src/driver.py:146-155)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    import convtasnet_oracle as O
    from speech_amd.trainer import allreduce_flat, shard_batch

    cfg = dict(O.CONFIG_SMALL, sep_num_layers=2)
    sd = cases.e2e_state(cfg)
    mixture, sources = cases.e2e_batch(cfg, 4, 4000, 0)
    mix_r, src_r = shard_batch(mixture, sources, rank, world)
    _, perm, _, grads = O.loss_and_grads(mix_r, src_r, sd, cfg)          # gradient of the shard-mean loss
    flat = torch.cat([grads[k].reshape(-1) for k in sd])
    flat_b = flat.clone()
    allreduce_flat(flat, 1)
    allreduce_flat(flat_b, 3)                                             # bucketed: identical result
    assert torch.equal(flat, flat_b)
    flat.mul_(1.0 / world)
    # clip after the reduction, then Adam: every rank must end with identical parameters
    glist = [flat]
    O.clip_grad_norm_(glist, 5.0)
    off, g2 = 0, {}
    for k, v in sd.items():
        g2[k] = flat[off:off + v.numel()].view(v.shape)
        off += v.numel()
    opt = O.AdamState(sd)
    opt.step(sd, g2)
    torch.save({"params": sd, "perm": perm}, os.path.join(out_dir, "rank{}.pt".format(rank)))
    dist.destroy_process_group()


def test_two_rank_data_parallel_equals_global_batch(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cases
    import convtasnet_oracle as O
    cfg = dict(O.CONFIG_SMALL, sep_num_layers=2)
    sd = cases.e2e_state(cfg)
    mixture, sources = cases.e2e_batch(cfg, 4, 4000, 0)
    opt = O.AdamState(sd)
    _, perm = O.train_step(mixture, sources, sd, cfg, opt, max_norm=5.0)     # single process, global batch
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"), weights_only=True)
    assert torch.equal(torch.cat([r0["perm"], r1["perm"]]), perm)
    for k in sd:
        assert torch.equal(r0["params"][k], r1["params"][k]), "ranks diverged on " + k
        d = (r0["params"][k] - sd[k]).abs().max().item()
        assert d <= 2e-6, "{}: data-parallel update differs from the global-batch update by {}".format(k, d)
