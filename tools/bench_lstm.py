"""Times the persistent LSTM kernels alone at the dual-path blocks' shapes (hidden 128, two directions)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
dev = torch.device("cuda:0")
for (N, T) in ((2052, 250), (1000, 513), (4096, 250)):
    H, D = 128, 2
    xproj = torch.randn(N, T, D, 4 * H, device=dev) * 0.5
    whh = torch.randn(D, 4 * H, H, device=dev) * 0.1
    dh = torch.randn(N, T, D, H, device=dev)
    hout, save = ops.lstm_fwd(xproj, whh)
    def timeit(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    tf = timeit(lambda: ops.lstm_fwd(xproj, whh))
    tb = timeit(lambda: ops.lstm_bwd(dh, save, whh))
    print("N=%d T=%d: fwd %.2f ms (%.2f us/step)  bwd %.2f ms (%.2f us/step)" % (N, T, tf, tf / T * 1e3, tb, tb / T * 1e3), flush=True)
