import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from speech_amd.functional import DepthwiseFn
dev = torch.device("cuda:0")
T, d, C, B, P = 3, 1, 1, 1, 3
x = torch.tensor([[[1., 2., 3.]]], dtype=torch.float64)
w = torch.tensor([[[100., 10., 1.]]], dtype=torch.float64)
b = torch.zeros(C, dtype=torch.float64)
ref = F.conv1d(F.pad(x, (d, d)), w, b, dilation=d, groups=C)
y = DepthwiseFn.apply(x.float().to(dev), w.float().to(dev), b.float().to(dev), d, d, d)
print("ref", ref.flatten().tolist())
print("got", y.flatten().tolist())
