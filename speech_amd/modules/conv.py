"""Depthwise separable convolution with the reference's interface (src/modules/conv.py:13-29):
depthwise Conv1d -> pointwise Conv1d, nothing in between, on the HIP depthwise and MFMA kernels."""
import torch.nn as nn

from ..functional import Conv1x1Fn, DepthwiseFn, DepthwiseGenericFn


class DepthwiseSeparableConv1d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=None, padding=0, dilation=1, bias=True):
        super().__init__()
        if stride is None:
            stride = kernel_size
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.padding = padding
        # parameter containers (state_dict keys as in the reference); their forwards are not used
        self.depthwise_conv1d = nn.Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride,
                                          padding=padding, dilation=dilation, groups=in_channels, bias=bias)
        self.pointwise_conv1d = nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1, bias=bias)

    def forward(self, input):
        pad = self.padding if isinstance(self.padding, int) else self.padding[0]
        stride = self.stride if isinstance(self.stride, int) else self.stride[0]
        span = (self.kernel_size - 1) * self.dilation
        dw = self.depthwise_conv1d
        if stride == 1 and 2 * pad == span:
            # 'same' length at stride 1: the tiled kernel of the TCN layers
            x = DepthwiseFn.apply(input, dw.weight, dw.bias, self.dilation, pad, pad)
        else:
            x = DepthwiseGenericFn.apply(input, dw.weight, dw.bias, stride, self.dilation, pad)
        pw = self.pointwise_conv1d
        return Conv1x1Fn.apply(x, pw.weight, pw.bias)
