"""CPU restatement of the reference's filterbank variants (Fourier, pseudo-inverse, gated) and of ConvTasNet's forward
pass with them.  TEST INFRASTRUCTURE ONLY: imported by tests/ (never by the speech_amd package); every function cites the
reference lines it follows.  Pinned by tests/golden/filterbank.npz, which tools/gen_golden_filterbank.py wrote by running
the reference's own src/models/filterbank.py.  Dtype-generic (float64 for tight comparisons).
"""
import torch
import torch.nn.functional as F

import convtasnet_oracle as O

EPS = 1e-12
FOURIER = ("Fourier", "trainableFourier", "trainableFourierTrainablePhase")


def dft_rows(frequency, time_seq, phase, sign, mirrored):
    """Real and imaginary rows cos / sin(sign * (omega n + phi)) (src/models/filterbank.py:49-54, :149-154); `mirrored`
    appends the interior bins in reverse order with the imaginary part negated (:57-61, :157-160)."""
    arg = frequency[:, None] * time_seq[None, :].to(frequency.dtype)
    if phase is not None:
        arg = arg + phase[:, None]
    re, im = torch.cos(sign * arg), torch.sin(sign * arg)
    if mirrored:
        idx = torch.arange(re.shape[0] - 2, 0, -1)
        re, im = torch.cat([re, re[idx]]), torch.cat([im, -im[idx]])
    return re, im


def fourier_encode(x, p, stride, onesided, return_complex):
    """FourierEncoder.forward: src/models/filterbank.py:33-73.  p: dict with frequency, time_seq, window[, phase]."""
    re, im = dft_rows(p["frequency"], p["time_seq"], p.get("phase"), -1.0, not onesided)
    re, im = (p["window"] * re)[:, None], (p["window"] * im)[:, None]                        # :62
    out_re, out_im = F.conv1d(x, re, stride=stride), F.conv1d(x, im, stride=stride)           # :63
    if return_complex:
        return torch.complex(out_re, out_im)                                                  # :64-67
    return torch.cat([out_re, out_im], dim=1)                                                 # :69-71


def _halves(z):
    if torch.is_complex(z):
        return z.real, z.imag
    return z[:, :z.shape[1] // 2], z[:, z.shape[1] // 2:]                                     # :146-147


def _synthesize(z, re, im, stride, onesided):
    z_re, z_im = _halves(z)
    if onesided:                                                                              # :164-168
        idx = torch.arange(z_re.shape[1] - 2, 0, -1)
        z_re, z_im = torch.cat([z_re, z_re[:, idx]], dim=1), torch.cat([z_im, -z_im[:, idx]], dim=1)
    return F.conv_transpose1d(z_re, re[:, None], stride=stride) - F.conv_transpose1d(z_im, im[:, None], stride=stride)


def fourier_decode(z, p, n_basis, stride, onesided):
    """FourierDecoder.forward: src/models/filterbank.py:127-172.  p: frequency, time_seq, optimal_window[, phase]."""
    re, im = dft_rows(p["frequency"], p["time_seq"], p.get("phase"), 1.0, True)
    re, im = p["optimal_window"] * re / n_basis, p["optimal_window"] * im / n_basis           # :161-162
    return _synthesize(z, re, im, stride, onesided)                                           # :170


def pinv_decode_trainable(z, enc_weight, stride):
    """PinvDecoder.forward with a trainable encoder: src/models/filterbank.py:279-282."""
    duplicate = enc_weight.shape[-1] // stride
    basis = torch.linalg.pinv(enc_weight[:, 0, :])                                            # (L, N)
    return F.conv_transpose1d(z, basis.t()[:, None, :] / duplicate, stride=stride)


def pinv_decode_fourier(z, p, n_basis, stride):
    """PinvDecoder.forward with a two-sided real-output Fourier encoder: src/models/filterbank.py:283-309 (the
    encoder's own window, not the optimal one)."""
    re, im = dft_rows(p["frequency"], p["time_seq"], p.get("phase"), 1.0, True)
    re, im = p["window"] * re / n_basis, p["window"] * im / n_basis                           # :307-308
    return _synthesize(z, re, im, stride, False)


def gated_encode(x, weight_U, weight_V, stride, eps=EPS):
    """GatedEncoder.forward: src/models/filterbank.py:337-346."""
    x = x / (torch.linalg.norm(x, dim=2, keepdim=True) + eps)
    return F.relu(F.conv1d(x, weight_U, stride=stride)) * torch.sigmoid(F.conv1d(x, weight_V, stride=stride))


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def conv_tasnet_forward(x, sd, cfg):
    """ConvTasNet.extract_latent (src/models/conv_tasnet.py:121-171) with the filterbank chosen as
    src/utils/filterbank.py:5-46 does from cfg['enc_basis'] / cfg['dec_basis']; 4-D input (B, 1, n_mics, T) is viewed
    as n_mics input channels (:138-141), a complex latent is masked in magnitude with its phase kept (:152-156)."""
    n_mics = None
    if x.dim() == 4:
        n_mics = x.shape[2]
        x = x.reshape(x.shape[0], n_mics, x.shape[3])
    B, _, T = x.shape
    L = cfg["kernel_size"]
    S = cfg.get("stride") or L // 2
    n_src, N = cfg["n_sources"], cfg["n_basis"]
    eb, db = cfg["enc_basis"], cfg["dec_basis"]
    onesided, cplx = bool(cfg.get("enc_onesided")), bool(cfg.get("enc_return_complex"))
    padding = (S - (T - L) % S) % S
    pad_l = padding // 2
    pad_r = padding - pad_l
    xp = F.pad(x, (pad_l, pad_r))
    if eb in FOURIER or db in FOURIER:      # compute_valid_basis, src/utils/filterbank.py:52-67
        if onesided:
            n_dft = 2 * (N - 1) if cplx else 2 * (N // 2 - 1)
        else:
            n_dft = N if cplx else N // 2
    if eb == "trainable":
        w = O.encoder(xp, sd["encoder.conv1d.weight"], S, None if db == "pinv" else cfg.get("enc_nonlinear"))
    elif eb in FOURIER:
        w = fourier_encode(xp, sub(sd, "encoder."), S, onesided, cplx)
    elif eb == "trainableGated":
        w = gated_encode(xp, sd["encoder.conv1d_U.weight"], sd["encoder.conv1d_V.weight"], S, cfg.get("eps", EPS))
    if torch.is_complex(w):
        amplitude, phase = torch.abs(w), torch.angle(w)
        mask = O.separator(amplitude, sd, cfg)
        w_hat = amplitude.unsqueeze(1) * mask * torch.exp(1j * phase.unsqueeze(1))
    else:
        w_hat = w.unsqueeze(1) * O.separator(w, sd, cfg)
    w_hat = w_hat.reshape(B * n_src, N, -1)
    if db == "trainable":
        x_hat = O.decoder(w_hat, sd["decoder.conv_transpose1d.weight"], S)
    elif db in FOURIER:
        x_hat = fourier_decode(w_hat, sub(sd, "decoder."), n_dft, S, onesided)
    elif db == "pinv" and eb == "trainable":
        x_hat = pinv_decode_trainable(w_hat, sd["encoder.conv1d.weight"], S)
    else:
        x_hat = pinv_decode_fourier(w_hat, sub(sd, "encoder."), n_dft, S)
    x_hat = x_hat.view(B, n_src, -1) if n_mics is None else x_hat.view(B, n_src, n_mics, -1)
    return F.pad(x_hat, (-pad_l, -pad_r))
