"""Generate tests/golden/dsconv.npz by running the REFERENCE's generic DepthwiseSeparableConv1d (src/modules/conv.py)
on CPU over the geometries of tests/dsconv_cases.py: output, input gradient and parameter gradients for a fixed probe.
Runs only in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_dsconv.py
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
sys.path.insert(1, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from modules.conv import DepthwiseSeparableConv1d  # noqa: E402  (reference)
from dsconv_cases import CASES  # noqa: E402

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "speech_amd", "utils", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


def main():
    d = {"meta": np.array("reference run on torch {} CPU fp32 (reference pins torch 1.10)".format(torch.__version__))}
    for i, (key, (ci, co, K, stride, pad, dil, bias, T)) in enumerate(CASES.items()):
        m = DepthwiseSeparableConv1d(ci, co, K, stride=stride, padding=pad, dilation=dil, bias=bias)
        synth.fill_parameters(m, salt=9)
        x = synth.normal((2, ci, T), 400 + i, std=0.5).requires_grad_(True)
        y = m(x)
        (y * synth.normal(tuple(y.shape), 450 + i)).sum().backward()
        d[key + "_y"] = y.detach().numpy()
        d[key + "_dx"] = x.grad.numpy()
        for name, p in m.named_parameters():
            d[key + "_g_" + name] = p.grad.numpy()
    out = os.path.join(ROOT, "tests", "golden", "dsconv.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
