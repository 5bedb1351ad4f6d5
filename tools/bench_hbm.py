"""Achievable HBM bandwidth on this GPU with PyTorch's own streaming kernels (copy = read + write, fill = write only,
sum = read only), as the practical ceiling beside the 8 TB/s vendor peak. usage: [MB]"""
import sys
import torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = mb * 1024 * 1024 // 4
dev = torch.device("cuda:0")
x = torch.randn(n, device=dev)
y = torch.empty_like(x)


def timed(fn, nbytes, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    return "%.1f us, %.2f TB/s" % (us, nbytes / us * 1e-6)


print("buffer %d MB" % mb)
print("copy  (read + write):", timed(lambda: y.copy_(x), 2 * n * 4))
print("fill  (write):       ", timed(lambda: y.fill_(1.0), n * 4))
print("sum   (read):        ", timed(lambda: x.sum(), n * 4))
print("axpy  (2 read + write):", timed(lambda: torch.add(x, y, alpha=0.5, out=y), 3 * n * 4))
