"""Case definitions of tests/golden/filterbank.npz, shared by tools/gen_golden_filterbank.py (which runs the reference on
them) and the tests (which run the oracle and the HIP path on them)."""
import torch

T, L, S = 400, 16, 8
FRAMES = (T - L) // S + 1
N_DFT = 16          # transform length of the stand-alone Fourier cases: 9 one-sided bins, 16 two-sided
N_PINV = 24         # basis count of the trainable / gated stand-alone cases

FOURIER_ENCODERS = {
    "onesided_complex": dict(onesided=True, return_complex=True),
    "onesided_real": dict(onesided=True, return_complex=False, trainable=True),
    "twosided_complex": dict(onesided=False, return_complex=True, trainable=True, trainable_phase=True),
    "twosided_real": dict(onesided=False, return_complex=False, trainable=True, trainable_phase=True, window_fn="hamming"),
}
FOURIER_DECODERS = {   # constructor arguments, complex input?
    "onesided_real": (dict(onesided=True, trainable=True), False),
    "onesided_complex": (dict(onesided=True, trainable=True, trainable_phase=True), True),
    "twosided_real": (dict(onesided=False, trainable=True, trainable_phase=True, window_fn="hamming"), False),
}

E2E_T = 2000
E2E_BASE = dict(n_basis=64, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=32,
                sep_bottleneck_channels=16, sep_skip_channels=16, sep_kernel_size=3, sep_num_blocks=1,
                sep_num_layers=2, mask_nonlinear="sigmoid", causal=False, n_sources=2, eps=1e-12)
E2E = {
    "fourier": dict(enc_basis="trainableFourierTrainablePhase", dec_basis="trainableFourier", enc_onesided=True,
                    enc_return_complex=False, window_fn="hann"),
    "pinv": dict(enc_basis="trainable", dec_basis="pinv"),
    "gated": dict(enc_basis="trainableGated", dec_basis="trainable"),
    "nondilated": dict(enc_basis="trainable", dec_basis="trainable", dilated=False, sep_num_layers=3),
    "nondilated_causal": dict(enc_basis="trainable", dec_basis="trainable", dilated=False, causal=True, enc_nonlinear="relu"),
    "multimic": dict(enc_basis="trainable", dec_basis="trainable", in_channels=2, enc_nonlinear="relu"),
    "fourier_complex": dict(enc_basis="Fourier", dec_basis="Fourier", enc_onesided=True, enc_return_complex=True,
                            window_fn="hann", n_basis=33),
    "fourier_pinv": dict(enc_basis="trainableFourier", dec_basis="pinv", enc_onesided=False, enc_return_complex=False,
                         window_fn="hann"),
}
# dense dilated heads instead of the depthwise-separable convolution (src/models/tdcn.py:102-105, :136-142);
# fixture tests/golden/nonseparable.npz, written by tools/gen_golden_nonseparable.py
E2E_NONSEPARABLE = {
    "nonseparable": dict(enc_basis="trainable", dec_basis="trainable", separable=False, sep_num_layers=3),
    "nonseparable_causal": dict(enc_basis="trainable", dec_basis="trainable", separable=False, causal=True, n_sources=3,
                                enc_nonlinear="relu", sep_num_blocks=2),
}
FOURIER_LEAVES = ("frequency", "time_seq", "window", "optimal_window", "phase")


def perturb(module, synth, salt=0):
    """Move the Fourier parameters off their analytic defaults so that every gradient term is exercised."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.split(".")[-1]
            seed = synth.name_seed(name, 40 + salt)
            if leaf == "frequency":
                p.add_(0.01 * synth.uniform(p.shape, seed))
            elif leaf in ("window", "optimal_window"):
                p.mul_(1.0 + 0.05 * synth.uniform(p.shape, seed))
            elif leaf == "phase":
                p.copy_(0.3 * synth.uniform(p.shape, seed))


def spectrum(synth, onesided, complex_in):
    bins = N_DFT // 2 + 1 if onesided else N_DFT
    z = synth.normal((2, 2 * bins, FRAMES), 305 + bins, std=0.5)
    return torch.complex(z[:, :bins].contiguous(), z[:, bins:].contiguous()) if complex_in else z


def fill_model(model, synth):
    """Hashed values for every convolution / norm / PReLU parameter, perturbed defaults for the Fourier ones."""
    state = {k: v for k, v in model.state_dict().items() if k.split(".")[-1] not in FOURIER_LEAVES}
    synth.fill_parameters(state, salt=2)
    perturb(model, synth, salt=1)


def e2e_batch(synth, cfg):
    """Mixture and sources of an end-to-end case: (2, 1, T) / (2, n_sources, T), or with a microphone axis
    (2, 1, n_mics, T) / (2, n_sources, n_mics, T) when the model takes several input channels."""
    n_mics = cfg.get("in_channels") or 1
    if n_mics == 1:
        return synth.make_batch(2, cfg["n_sources"], E2E_T, seed=321)
    sources = synth.normal((2, cfg["n_sources"], n_mics, E2E_T), 322, std=0.1)
    return sources.sum(dim=1, keepdim=True), sources
