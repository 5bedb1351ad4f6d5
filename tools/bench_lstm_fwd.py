"""Times ctn_lstm_fwd alone at the dual-path blocks' two shapes (hidden 128, two directions)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from speech_amd import ops
dev = torch.device("cuda:0")
out = []
for (N, T) in ((2052, 250), (1000, 513), (2048, 250)):
    H, D = 128, 2
    xproj = torch.randn(N, T, D, 4 * H, device=dev) * 0.5
    whh = torch.randn(D, 4 * H, H, device=dev) * 0.1
    ops.lstm_fwd(xproj, whh); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.lstm_fwd(xproj, whh)
    e1.record(); torch.cuda.synchronize()
    out.append("N=%d T=%d %.2f us/step" % (N, T, e0.elapsed_time(e1) / 5 / T * 1e3))
print(os.environ.get("CTN_LIB", "default"), " | ".join(out), flush=True)
