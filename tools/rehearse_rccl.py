"""One rank, RCCL backend: the pieces of the multi-GPU step that a one-GPU box can exercise -- communicator set-up with
device_id, weight broadcast, HIP-graph capture beside the process group's watchdog thread, bucketed all-reduce of the
flat gradient buffer on the compute stream, barrier.  Prints the loss trajectory with and without the collective."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
print("backend", dist.get_backend(), flush=True)

import convtasnet_oracle as O
from bench import build_model
from speech_amd.criterion.pit import PIT1d
from speech_amd.criterion.sdr import NegSISDR
from speech_amd.trainer import FlatTrainer, allreduce_flat
from speech_amd.utils import synth


def run(collective):
    model = build_model(O.CONFIG_SMALL, device)
    tr = FlatTrainer(model, PIT1d(NegSISDR(), 2), n_streams=2, use_graph=True, n_buckets=4)
    dist.broadcast(tr.flat_p, src=0)
    out = []
    for i in range(6):
        mix, src = synth.make_batch(4, 2, 8000, seed=i)
        loss, _ = tr.forward_backward(mix.to(device), src.to(device))
        if collective:
            allreduce_flat(tr.flat_g, 4)
        tr.optimizer_step()
        out.append(loss.item())
    dist.barrier()
    torch.cuda.synchronize()
    return out, tr.use_graph


a, ga = run(True)
b, gb = run(False)
print("with all-reduce   ", ["%.5f" % v for v in a], "graph" if ga else "eager")
print("without all-reduce", ["%.5f" % v for v in b], "graph" if gb else "eager")
assert ga and gb, "graph capture was refused"
assert max(abs(x - y) for x, y in zip(a, b)) < 1e-3 * max(abs(x) for x in b)
dist.destroy_process_group()
print("ok")
