"""Trainable encoder / decoder with the reference's interface and state_dict keys
(src/models/filterbank.py:205-251) on the HIP encoder / overlap-add kernels."""
import torch.nn as nn

from ..functional import EncoderFn, DecoderFn


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
        return EncoderFn.apply(input, self.conv1d.weight, self.stride, self.nonlinear)

    def get_basis(self):
        return self.conv1d.weight


class Decoder(nn.Module):
    def __init__(self, n_basis, out_channels, kernel_size=16, stride=8):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        if out_channels != 1:
            raise NotImplementedError("HIP decoder supports out_channels == 1")
        # parameter container: state_dict key `conv_transpose1d.weight` (N, 1, L)
        self.conv_transpose1d = nn.ConvTranspose1d(n_basis, out_channels, kernel_size=kernel_size, stride=stride,
                                                   bias=False)

    def forward(self, input):
        return DecoderFn.apply(input, self.conv_transpose1d.weight, self.stride)

    def get_basis(self):
        return self.conv_transpose1d.weight
