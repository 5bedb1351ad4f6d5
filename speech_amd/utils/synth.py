"""Counter-based synthetic data and parameter filler.

Values depend only on (seed, element index), never on a library RNG, so the
parity tests, the golden-vector generator and the benchmark regenerate the
same tensors on any machine without shipping them.
"""
import zlib

import numpy as np
import torch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    """splitmix64 finaliser on a uint64 array."""
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def uniform01(n, seed):
    """n doubles in [0, 1), element i a pure function of (seed, i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = _mix((idx + np.uint64(1)) * _GOLD + np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * _M2)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(shape, seed, lo=-1.0, hi=1.0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(n, seed) * (hi - lo) + lo
    return torch.from_numpy(u.reshape(shape)).to(dtype)


def normal(shape, seed, std=1.0, dtype=torch.float32):
    """Box-Muller on two hashed uniforms."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(n, 2 * seed + 1)
    u2 = uniform01(n, 2 * seed + 2)
    g = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((std * g).reshape(shape)).to(dtype)


def name_seed(name, salt=0):
    return (zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF


def fill_parameters(module_or_state, salt=0):
    """Overwrite every parameter with hashed values keyed by its state_dict name.

    Convolution weights/biases get U(+-1/sqrt(fan_in)) (the scale of the stock
    initialiser), norm scales 1 + 0.2u, norm shifts 0.2u, PReLU slopes
    0.25 + 0.1u, so that every term of every backward formula is exercised.
    Accepts an nn.Module (filled in place) or a dict name -> tensor.
    """
    state = module_or_state.state_dict() if hasattr(module_or_state, "state_dict") else module_or_state
    with torch.no_grad():
        for name, p in state.items():
            seed = name_seed(name, salt)
            leaf = name.split(".")[-1]
            owner = name.split(".")[-2] if "." in name else ""
            if "nonlinear1d" in owner or owner == "prelu":
                v = 0.25 + 0.1 * uniform(p.shape, seed)
            elif leaf == "gamma" or (owner == "norm" and leaf == "weight"):
                v = 1.0 + 0.2 * uniform(p.shape, seed)
            elif leaf == "beta" or (owner == "norm" and leaf == "bias"):
                v = 0.2 * uniform(p.shape, seed)
            elif leaf == "weight":
                fan_in = int(np.prod(p.shape[1:])) if p.dim() > 1 else 1
                v = uniform(p.shape, seed) / np.sqrt(max(fan_in, 1))
            else:  # conv bias
                v = 0.1 * uniform(p.shape, seed)
            p.copy_(v.to(p.dtype))
    return module_or_state


def make_batch(batch, n_sources, T, seed, std=0.1):
    """sources ~ std * N(0, 1), mixture = sum of sources (the benchmark workload)."""
    sources = normal((batch, n_sources, T), seed, std=std)
    mixture = sources.sum(dim=1, keepdim=True)
    return mixture, sources
