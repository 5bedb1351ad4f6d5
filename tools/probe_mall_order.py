"""Does a consumer gain from reading its producer's output last-written-first?  Kernel A copies src -> buf (ascending
chunks); kernel B copies buf -> dst in ascending or descending chunk order; B's duration for buffers of 64 .. 512 MiB.
Second table: B after a kernel that READ buf (ascending)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def run(a, y, rev):
    _lib.check(lib.ctn_probe_stream(a.data_ptr(), y.data_ptr(), a.numel(), rev, st), "probe")


for mode in ("after a write of buf", "after a read of buf"):
    print(mode)
    for mb in (64, 128, 192, 256, 320, 384, 512, 786):
        n = mb * 1024 * 1024 // 4
        src = torch.randn(n, device=dev); buf = torch.empty_like(src); dst = torch.empty_like(src); dst2 = torch.empty_like(src)
        flush = torch.empty(1024 * 1024 * 1024 // 4, device=dev)
        res = []
        for rev in (0, 1):
            ts = []
            for rep in range(5):
                flush.zero_()                            # 1 GiB of other traffic: evicts everything
                if mode.startswith("after a write"):
                    run(src, buf, 0)
                else:
                    run(buf, dst2, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(buf, dst, rev); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res.append(sorted(ts)[len(ts) // 2])
        print("  %4d MiB: ascending %7.1f us (%5.2f TB/s)   descending %7.1f us (%5.2f TB/s)   ratio %.2f" %
              (mb, res[0], 2 * n * 4 / res[0] / 1e6, res[1], 2 * n * 4 / res[1] / 1e6, res[0] / res[1]))
        del src, buf, dst, dst2, flush
