"""Encoders / decoders with the reference's interface and state_dict keys (src/models/filterbank.py:12-346) on the HIP
encoder / overlap-add kernels: the trainable pair of the hot path, and the Fourier, pseudo-inverse and gated variants,
whose (small) bases are assembled by tensor operations and whose convolutions run on the same two kernels."""
import math

import torch
import torch.nn as nn

from ..functional import EncoderFn, DecoderFn
from ..utils.audio import build_window, build_optimal_window

EPS = 1e-12


def _analysis(input, weight, stride, relu):
    """Conv1d(in_channels, N, L, stride, bias=False) on the one-channel HIP encoder kernel: a sum over the input
    channels (microphones), the ReLU after the sum."""
    if input.size(1) == 1:
        return EncoderFn.apply(input, weight, stride, relu)
    x = EncoderFn.apply(input[:, 0:1], weight[:, 0:1].contiguous(), stride, False)
    for c in range(1, input.size(1)):
        x = x + EncoderFn.apply(input[:, c:c + 1], weight[:, c:c + 1].contiguous(), stride, False)
    return torch.relu(x) if relu else x


class Encoder(nn.Module):
    def __init__(self, in_channels, n_basis, kernel_size=16, stride=8, nonlinear=None):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        # parameter container: state_dict key `conv1d.weight` (N, in_channels, L)
        self.conv1d = nn.Conv1d(in_channels, n_basis, kernel_size=kernel_size, stride=stride, bias=False)
        if nonlinear is not None:
            if nonlinear == 'relu':
                self.nonlinear1d = nn.ReLU()
            else:
                raise NotImplementedError("Not support {}".format(nonlinear))
            self.nonlinear = True
        else:
            self.nonlinear = False

    def forward(self, input):
        return _analysis(input, self.conv1d.weight, self.stride, self.nonlinear)

    def get_basis(self):
        return self.conv1d.weight


class Decoder(nn.Module):
    def __init__(self, n_basis, out_channels, kernel_size=16, stride=8):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        # parameter container: state_dict key `conv_transpose1d.weight` (N, 1, L)
        self.conv_transpose1d = nn.ConvTranspose1d(n_basis, out_channels, kernel_size=kernel_size, stride=stride,
                                                   bias=False)

    def forward(self, input):
        weight = self.conv_transpose1d.weight                      # (N, out_channels, L)
        if weight.size(1) == 1:
            return DecoderFn.apply(input, weight, self.stride)
        # one overlap-add pass per output channel (microphone)
        return torch.cat([DecoderFn.apply(input, weight[:, c:c + 1].contiguous(), self.stride)
                          for c in range(weight.size(1))], dim=1)

    def get_basis(self):
        return self.conv_transpose1d.weight


def _fourier_basis(omega, n, phase, sign, both_halves):
    """cos and sin of sign * (omega_k * n + phase_k), (n_bins, kernel_size) each.  both_halves appends the mirrored
    interior bins (real part repeated, imaginary part negated): the conjugate half of a full spectrum."""
    arg = omega.unsqueeze(dim=1) * n.unsqueeze(dim=0)
    if phase is not None:
        arg = arg + phase.unsqueeze(dim=1)
    re, im = torch.cos(sign * arg), torch.sin(sign * arg)
    if both_halves:
        re = torch.cat([re, torch.flip(re[1:-1], dims=(0,))], dim=0)
        im = torch.cat([im, - torch.flip(im[1:-1], dims=(0,))], dim=0)
    return re, im


def _split_complex(input):
    if torch.is_complex(input):
        return input.real, input.imag
    n_bins = input.size(1)
    return torch.split(input, [n_bins // 2, n_bins // 2], dim=1)


class FourierEncoder(nn.Module):
    """Windowed DFT analysis as a strided convolution (src/models/filterbank.py:12-103): the basis is built from the
    `frequency`, `window` (and optional `phase`) parameters on every call, the convolution runs on the HIP encoder
    kernel with the real rows stacked on the imaginary rows."""

    def __init__(self, n_basis, kernel_size, stride=None, window_fn='hann', trainable=False, trainable_phase=False,
                 onesided=True, return_complex=True):
        super().__init__()
        self.n_basis = n_basis
        self.kernel_size, self.stride = kernel_size, stride
        self.trainable, self.trainable_phase = trainable, trainable_phase
        self.onesided, self.return_complex = onesided, return_complex
        omega = 2 * math.pi * torch.arange(n_basis // 2 + 1) / n_basis
        self.frequency = nn.Parameter(omega, requires_grad=trainable)
        self.time_seq = nn.Parameter(torch.arange(kernel_size), requires_grad=False)
        self.window = nn.Parameter(build_window(kernel_size, window_fn=window_fn))
        if self.trainable_phase:
            self.phase = nn.Parameter(torch.zeros(n_basis // 2 + 1), requires_grad=True)

    def get_basis(self):
        """(2 * n_bins, kernel_size): windowed real rows, then windowed imaginary rows."""
        re, im = _fourier_basis(self.frequency, self.time_seq, self.phase if self.trainable_phase else None, -1.0,
                                not self.onesided)
        return torch.cat([self.window * re, self.window * im], dim=0)

    def forward(self, input):
        """
        Args:
            input (batch_size, 1, T)
        Returns:
            complex (batch_size, n_bins, n_frames) if return_complex else real (batch_size, 2 * n_bins, n_frames),
            n_bins = n_basis // 2 + 1 if onesided else n_basis
        """
        basis = self.get_basis()
        output = EncoderFn.apply(input, basis.unsqueeze(dim=1), self.stride, False)
        if self.return_complex:
            n_bins = basis.size(0) // 2
            output = torch.complex(output[:, :n_bins], output[:, n_bins:])
        return output

    def extra_repr(self):
        s = "{n_basis}, kernel_size={kernel_size}, stride={stride}, trainable={trainable}, onesided={onesided}, return_complex={return_complex}"
        if self.trainable_phase:
            s += ", trainable_phase={trainable_phase}"
        return s.format(**self.__dict__)


def _synthesis(input, re, im, stride, onesided):
    """conv_transpose(real part, re) - conv_transpose(imaginary part, im) as ONE overlap-add pass of the HIP decoder
    kernel over the stacked channels.  A one-sided input is first completed with its mirrored interior bins."""
    input_real, input_imag = _split_complex(input)
    if onesided:
        input_real = torch.cat([input_real, torch.flip(input_real[:, 1:-1], dims=(1,))], dim=1)
        input_imag = torch.cat([input_imag, - torch.flip(input_imag[:, 1:-1], dims=(1,))], dim=1)
    stacked = torch.cat([input_real, input_imag], dim=1)
    return DecoderFn.apply(stacked, torch.cat([re, - im], dim=0).unsqueeze(dim=1), stride)


class FourierDecoder(nn.Module):
    """Inverse windowed DFT as a transposed convolution with the optimal synthesis window
    (src/models/filterbank.py:105-203)."""

    def __init__(self, n_basis, kernel_size, stride=None, window_fn='hann', trainable=False, trainable_phase=False,
                 onesided=True):
        super().__init__()
        self.n_basis = n_basis
        self.kernel_size, self.stride = kernel_size, stride
        self.trainable, self.trainable_phase = trainable, trainable_phase
        self.onesided = onesided
        omega = 2 * math.pi * torch.arange(n_basis // 2 + 1) / n_basis
        window = build_window(kernel_size, window_fn=window_fn)
        self.frequency = nn.Parameter(omega, requires_grad=trainable)
        self.time_seq = nn.Parameter(torch.arange(kernel_size), requires_grad=False)
        self.optimal_window = nn.Parameter(build_optimal_window(window, hop_length=stride))
        if self.trainable_phase:
            self.phase = nn.Parameter(torch.zeros(n_basis // 2 + 1), requires_grad=True)

    def _basis(self, both_halves):
        re, im = _fourier_basis(self.frequency, self.time_seq, self.phase if self.trainable_phase else None, 1.0,
                                both_halves)
        return self.optimal_window * re / self.n_basis, self.optimal_window * im / self.n_basis

    def forward(self, input):
        """
        Args:
            input: complex (batch_size, n_bins, n_frames) or real (batch_size, 2 * n_bins, n_frames)
        Returns:
            output (batch_size, 1, T)
        """
        re, im = self._basis(True)
        return _synthesis(input, re, im, self.stride, self.onesided)

    def get_basis(self):
        re, im = self._basis(not self.onesided)
        return torch.cat([re, im], dim=0)

    def extra_repr(self):
        s = "{n_basis}, kernel_size={kernel_size}, stride={stride}, trainable={trainable}, onesided={onesided}"
        if self.trainable_phase:
            s += ", trainable_phase={trainable_phase}"
        return s.format(**self.__dict__)


class PinvDecoder(nn.Module):
    """Decoder tied to an encoder (src/models/filterbank.py:253-323): the pseudo-inverse of a trainable basis (divided
    by the number of frames that overlap a sample), or the inverse transform of a two-sided Fourier encoder."""

    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder
        self.kernel_size, self.stride = encoder.kernel_size, encoder.stride
        if isinstance(encoder, Encoder):
            if encoder.nonlinear:
                raise ValueError("Not support pseudo inverse of 'Conv1d + nonlinear'.")
            self.weight = encoder.conv1d.weight
            n_rows, _, n_columns = self.weight.size()
            if n_rows < n_columns:
                raise ValueError("Cannot compute the left inverse of encoder's weight. In encoder, `out_channels` must be equal to or greater than `kernel_size`.")
        elif isinstance(encoder, FourierEncoder):
            if encoder.onesided or encoder.return_complex:
                raise ValueError("Both encoder.onesided and encoder.return_complex are expected to be False.")
        else:
            raise TypeError("Invalid encoder is given.")

    def get_basis(self):
        duplicate = self.kernel_size // self.stride
        weight = self.weight.permute(1, 0, 2)                       # (1, N, L)
        return torch.pinverse(weight).permute(2, 0, 1).contiguous() / duplicate

    def forward(self, input):
        encoder = self.encoder
        if isinstance(encoder, Encoder):
            return DecoderFn.apply(input, self.get_basis(), self.stride)
        re, im = _fourier_basis(encoder.frequency, encoder.time_seq, encoder.phase if encoder.trainable_phase else None,
                                1.0, True)
        re, im = encoder.window * re / encoder.n_basis, encoder.window * im / encoder.n_basis
        return _synthesis(input, re, im, self.stride, False)


class GatedEncoder(nn.Module):
    """TasNet's gated encoder (src/models/filterbank.py:325-346): the input is scaled to unit norm, then
    relu(U x) * sigmoid(V x) with two trainable bases, both on the HIP encoder kernel."""

    def __init__(self, in_channels, n_basis, kernel_size=16, stride=8, eps=EPS):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self.eps = eps
        self.conv1d_U = nn.Conv1d(in_channels, n_basis, kernel_size=kernel_size, stride=stride, bias=False)
        self.conv1d_V = nn.Conv1d(in_channels, n_basis, kernel_size=kernel_size, stride=stride, bias=False)
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()

    def forward(self, input):
        norm = torch.linalg.norm(input, dim=2, keepdim=True)
        x = input / (norm + self.eps)
        x_U = _analysis(x, self.conv1d_U.weight, self.stride, True)
        x_V = _analysis(x, self.conv1d_V.weight, self.stride, False)
        return x_U * self.sigmoid(x_V)
