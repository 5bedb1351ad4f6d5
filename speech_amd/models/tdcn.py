"""Time-dilated convolutional network with the reference's interface and state_dict keys
(src/models/tdcn.py), running on the fused HIP kernels (see speech_amd/functional.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functional import ResidualBlockFn, LayerNormFn, Conv1x1Fn, DepthwiseFn, DepthwiseGenericFn
from ..utils.tasnet import choose_layer_norm

EPS = 1e-12


def _norm_params(norm):
    """(gamma, beta) of a GlobalLayerNorm or CumulativeLayerNorm1d."""
    if hasattr(norm, "gamma"):
        return norm.gamma, norm.beta
    return norm.norm.weight, norm.norm.bias


class TimeDilatedConvNet(nn.Module):
    """src/models/tdcn.py:13-41: num_blocks x num_layers residual layers, skip outputs summed.
    The very last layer has no residual head (:22-23, :58-59)."""

    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, num_blocks=3,
                 num_layers=10, dilated=True, separable=False, causal=True, nonlinear=None, norm=True, eps=EPS):
        super().__init__()
        self.num_blocks = num_blocks
        net = []
        for idx in range(num_blocks):
            net.append(TimeDilatedConvBlock1d(
                num_features, hidden_channels=hidden_channels, skip_channels=skip_channels, kernel_size=kernel_size,
                num_layers=num_layers, dilated=dilated, separable=separable, causal=causal, nonlinear=nonlinear,
                norm=norm, dual_head=(idx != num_blocks - 1), eps=eps))
        self.net = nn.Sequential(*net)

    def forward(self, input):
        x = input
        skip_connection = 0
        for idx in range(self.num_blocks):
            x, skip = self.net[idx](x)
            skip_connection = skip_connection + skip
        return skip_connection


class TimeDilatedConvBlock1d(nn.Module):
    """src/models/tdcn.py:43-75: dilation 2**idx per layer (:53)."""

    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, num_layers=10,
                 dilated=True, separable=False, causal=True, nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.num_layers = num_layers
        net = []
        for idx in range(num_layers):
            if dilated:
                dilation, stride = 2 ** idx, 1
            else:
                dilation, stride = 1, 2
            last = (not dual_head) and idx == num_layers - 1
            net.append(ResidualBlock1d(
                num_features, hidden_channels=hidden_channels, skip_channels=skip_channels, kernel_size=kernel_size,
                stride=stride, dilation=dilation, separable=separable, causal=causal, nonlinear=nonlinear, norm=norm,
                dual_head=not last, eps=eps))
        self.net = nn.Sequential(*net)

    def forward(self, input):
        x = input
        skip_connection = 0
        for idx in range(self.num_layers):
            x, skip = self.net[idx](x)
            skip_connection = skip_connection + skip
        return x, skip_connection


class ResidualBlock1d(nn.Module):
    """src/models/tdcn.py:77-147.  The HIP path implements the Conv-TasNet configuration of this block:
    stride 1 (dilated), separable, PReLU, norm."""

    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, stride=2, dilation=1,
                 separable=False, causal=True, nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.separable, self.causal = separable, causal
        self.norm = norm
        self.dual_head = dual_head
        self.eps = eps

        self.bottleneck_conv1d = nn.Conv1d(num_features, hidden_channels, kernel_size=1, stride=1)
        if nonlinear is not None:
            if nonlinear == 'prelu':
                self.nonlinear1d = nn.PReLU()
            else:
                raise ValueError("Not support {}".format(nonlinear))
            self.nonlinear = True
        else:
            self.nonlinear = False
        if norm:
            norm_name = 'cLN' if causal else 'gLN'
            self.norm1d = choose_layer_norm(norm_name, hidden_channels, causal=causal, eps=eps)
        if separable:
            self.separable_conv1d = DepthwiseSeparableConv1d(
                hidden_channels, num_features, skip_channels=skip_channels, kernel_size=kernel_size, stride=stride,
                dilation=dilation, causal=causal, nonlinear=nonlinear, norm=norm, dual_head=dual_head, eps=eps)
        else:
            # dense dilated heads (src/models/tdcn.py:102-105): these modules hold the parameters under the reference's
            # names; forward() runs every tap as a pointwise GEMM over a shifted view of the padded input
            if dual_head:
                self.output_conv1d = nn.Conv1d(hidden_channels, num_features, kernel_size=kernel_size, dilation=dilation)
            self.skip_conv1d = nn.Conv1d(hidden_channels, skip_channels, kernel_size=kernel_size, dilation=dilation)

    def hip_supported(self):
        return self.separable and self.nonlinear and self.norm and self.stride == 1

    @staticmethod
    def _dense_dilated(x_padded, conv, T_out):
        """nn.Conv1d(C_in, C_out, P, dilation=d) on an already padded input as P pointwise convolutions of shifted
        views, summed: y = b + sum_k W[:, :, k] . x[:, :, k d : k d + T_out] -- each a ctn_gemm_nn / ctn_gemm_nt pair."""
        P, d = conv.kernel_size[0], conv.dilation[0]
        y = None
        for k in range(P):
            tap = Conv1x1Fn.apply(x_padded[:, :, k * d:k * d + T_out], conv.weight[:, :, k:k + 1].contiguous(),
                                  conv.bias if k == 0 else None)
            y = tap if y is None else y + tap
        return y

    def parameter_list(self):
        """Parameters in the order speech_amd.functional.Layer expects."""
        s = self.separable_conv1d
        g1, be1 = _norm_params(self.norm1d)
        g2, be2 = _norm_params(s.norm1d)
        ps = [self.bottleneck_conv1d.weight, self.bottleneck_conv1d.bias, self.nonlinear1d.weight, g1, be1,
              s.depthwise_conv1d.weight, s.depthwise_conv1d.bias, s.nonlinear1d.weight, g2, be2]
        if self.dual_head:
            ps += [s.output_pointwise_conv1d.weight, s.output_pointwise_conv1d.bias]
        ps += [s.skip_pointwise_conv1d.weight, s.skip_pointwise_conv1d.bias]
        return ps

    def forward(self, input):
        if self.hip_supported():
            output, skip = ResidualBlockFn.apply(input, self.dilation, self.causal, self.dual_head, self.eps,
                                                 *self.parameter_list())
            return output, skip
        if not (self.nonlinear and self.norm):
            raise NotImplementedError("HIP path: ResidualBlock1d needs prelu and norm")
        if not self.separable and self.stride != 1:
            # the reference's dense heads take no stride (src/models/tdcn.py:103-105) while its padding assumes one
            # (:123): with stride 2 its own output no longer matches the residual in length
            raise NotImplementedError("non-separable residual layers exist for stride 1 (dilated=True) only")
        # strided (non-dilated) layers, src/models/tdcn.py:55-57: stage by stage on the same kernels; the padding keeps
        # the frame count (:123-132), so most of what the stride-2 depthwise convolution reads is zero padding
        kernel_size, stride, dilation = self.kernel_size, self.stride, self.dilation
        T_original = input.size(-1)
        b = self.bottleneck_conv1d
        x = Conv1x1Fn.apply(input, b.weight, b.bias)
        x = F.prelu(x, self.nonlinear1d.weight)
        g, be = _norm_params(self.norm1d)
        x = LayerNormFn.apply(x, g, be, self.eps, self.causal)
        padding = (T_original - 1) * stride - T_original + (kernel_size - 1) * dilation + 1
        padding_left = padding if self.causal else padding // 2
        x = F.pad(x, (padding_left, padding - padding_left))
        if self.separable:
            output, skip = self.separable_conv1d(x)
        else:                                          # src/models/tdcn.py:136-142
            output = self._dense_dilated(x, self.output_conv1d, T_original) if self.dual_head else None
            skip = self._dense_dilated(x, self.skip_conv1d, T_original)
        if output is not None:
            output = output + input
        return output, skip


class DepthwiseSeparableConv1d(nn.Module):
    """The dual-head block of src/models/tdcn.py:149-196: depthwise conv -> PReLU -> norm -> output and skip
    1x1 heads.  Inside ResidualBlock1d it only holds the parameters (the fused kernels run the arithmetic);
    called on its own it runs the same kernels unfused on an already padded input."""

    def __init__(self, in_channels, out_channels=256, skip_channels=256, kernel_size=3, stride=2, dilation=1,
                 causal=True, nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.dual_head = dual_head
        self.norm = norm
        self.eps = eps
        self.causal = causal
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation

        self.depthwise_conv1d = nn.Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride,
                                          dilation=dilation, groups=in_channels)
        if nonlinear is not None:
            if nonlinear == 'prelu':
                self.nonlinear1d = nn.PReLU()
            else:
                raise ValueError("Not support {}".format(nonlinear))
            self.nonlinear = True
        else:
            self.nonlinear = False
        if norm:
            norm_name = 'cLN' if causal else 'gLN'
            self.norm1d = choose_layer_norm(norm_name, in_channels, causal=causal, eps=eps)
        if dual_head:
            self.output_pointwise_conv1d = nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1)
        self.skip_pointwise_conv1d = nn.Conv1d(in_channels, skip_channels, kernel_size=1, stride=1)

    def forward(self, input):
        span = (self.kernel_size - 1) * self.dilation
        T_out = input.size(-1) - span
        dw = self.depthwise_conv1d
        if self.stride == 1:
            x = DepthwiseFn.apply(input, dw.weight, dw.bias, self.dilation, 0, span)[..., :T_out]
        else:
            x = DepthwiseGenericFn.apply(input, dw.weight, dw.bias, self.stride, self.dilation, 0)
        if self.nonlinear:
            x = F.prelu(x, self.nonlinear1d.weight)
        if self.norm:
            g, b = _norm_params(self.norm1d)
            x = LayerNormFn.apply(x, g, b, self.eps, self.causal)
        if self.dual_head:
            o = self.output_pointwise_conv1d
            output = Conv1x1Fn.apply(x, o.weight, o.bias)
        else:
            output = None
        s = self.skip_pointwise_conv1d
        skip = Conv1x1Fn.apply(x, s.weight, s.bias)
        return output, skip
