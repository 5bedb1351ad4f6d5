import sys; sys.path.insert(0, "/root/repo"); sys.argv=["x"]
import torch, bench
d = torch.device("cuda:0")
for _ in range(2):
    print("f32 bare", bench.mfma_bare_loop_rate(d), "bf16 bare", bench.mfma_bare_loop_rate(d, bf16=True))
