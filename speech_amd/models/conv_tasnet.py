"""Conv-TasNet with the reference's constructor, methods and state_dict keys
(src/models/conv_tasnet.py), running on hand-written gfx950 kernels.

forward() runs the whole network as one fused schedule (speech_amd.functional.ConvTasNetFn);
extract_latent() runs encoder, separator and decoder as separate modules so the latent is
available (and differentiable), exactly as the reference exposes it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functional import ConvTasNetFn, SeparatorFn, Layer, LayerNormFn, Conv1x1Fn
from ..utils.filterbank import choose_filterbank
from ..utils.model import choose_nonlinear
from ..utils.tasnet import choose_layer_norm, frame_padding, waveform_rows, decode_sources
from .tdcn import TimeDilatedConvNet, _norm_params

EPS = 1e-12


class ConvTasNet(nn.Module):
    def __init__(self,
                 n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None,
                 sep_hidden_channels=256, sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3,
                 sep_num_blocks=3, sep_num_layers=8,
                 dilated=True, separable=True,
                 sep_nonlinear='prelu', sep_norm=True, mask_nonlinear='sigmoid',
                 causal=True,
                 n_sources=2,
                 eps=EPS,
                 **kwargs):
        super().__init__()

        if stride is None:
            stride = kernel_size // 2
        assert kernel_size % stride == 0, "kernel_size is expected divisible by stride"

        # Encoder-decoder
        self.in_channels = kwargs.get('in_channels', 1)
        self.n_basis = n_basis
        self.kernel_size, self.stride = kernel_size, stride
        self.enc_basis, self.dec_basis = enc_basis, dec_basis

        if enc_basis == 'trainable' and not dec_basis == 'pinv':
            self.enc_nonlinear = kwargs['enc_nonlinear']
        else:
            self.enc_nonlinear = None

        if enc_basis in ['Fourier', 'trainableFourier', 'trainableFourierTrainablePhase'] or \
                dec_basis in ['Fourier', 'trainableFourier', 'trainableFourierTrainablePhase']:
            self.window_fn = kwargs['window_fn']
            self.enc_onesided, self.enc_return_complex = kwargs['enc_onesided'], kwargs['enc_return_complex']
        else:
            self.window_fn = None
            self.enc_onesided, self.enc_return_complex = None, None

        # Separator configuration
        self.sep_hidden_channels, self.sep_bottleneck_channels, self.sep_skip_channels = \
            sep_hidden_channels, sep_bottleneck_channels, sep_skip_channels
        self.sep_kernel_size = sep_kernel_size
        self.sep_num_blocks, self.sep_num_layers = sep_num_blocks, sep_num_layers
        self.dilated, self.separable, self.causal = dilated, separable, causal
        self.sep_nonlinear, self.sep_norm = sep_nonlinear, sep_norm
        self.mask_nonlinear = mask_nonlinear
        self.n_sources = n_sources
        self.eps = eps

        # Network configuration
        encoder, decoder = choose_filterbank(n_basis, kernel_size=kernel_size, stride=stride, enc_basis=enc_basis,
                                             dec_basis=dec_basis, **kwargs)
        self.encoder = encoder
        self.separator = Separator(
            n_basis, bottleneck_channels=sep_bottleneck_channels, hidden_channels=sep_hidden_channels,
            skip_channels=sep_skip_channels, kernel_size=sep_kernel_size, num_blocks=sep_num_blocks,
            num_layers=sep_num_layers, dilated=dilated, separable=separable, causal=causal, nonlinear=sep_nonlinear,
            norm=sep_norm, mask_nonlinear=mask_nonlinear, n_sources=n_sources, eps=eps)
        self.decoder = decoder

    # ------------------------------------------------------------------ fused path
    def _fused_ok(self):
        return (self.enc_basis == 'trainable' and self.dec_basis in ('trainable', 'pinv') and self.in_channels == 1
                and self.dilated and self.separable and self.sep_nonlinear == 'prelu' and self.sep_norm
                and self.mask_nonlinear == 'sigmoid' and self.kernel_size <= 32 and self.stride <= 16
                and (self.kernel_size - 1) // self.stride < 32 and self.n_sources <= (4 if self.kernel_size <= 16 else 3))

    def fused_config(self):
        return dict(n_basis=self.n_basis, kernel_size=self.kernel_size, stride=self.stride,
                    enc_nonlinear=self.enc_nonlinear, sep_num_blocks=self.sep_num_blocks,
                    sep_num_layers=self.sep_num_layers, causal=self.causal, n_sources=self.n_sources, eps=self.eps)

    def fused_parameters(self):
        """Parameters in the order speech_amd.functional.ModelParams expects."""
        sep = self.separator
        g0, be0 = _norm_params(sep.norm1d)
        ps = [self.encoder.conv1d.weight, g0, be0, sep.bottleneck_conv1d.weight, sep.bottleneck_conv1d.bias]
        ps += sep.layer_parameters()
        # a pseudo-inverse decoder hands the fused schedule its derived basis; autograd carries the gradient on to the
        # encoder weight through torch.pinverse
        dec_W = self.decoder.get_basis() if self.dec_basis == 'pinv' else self.decoder.conv_transpose1d.weight
        ps += [sep.prelu.weight, sep.mask_conv1d.weight, sep.mask_conv1d.bias, dec_W]
        return ps

    def forward(self, input):
        # the fused schedule does not produce the gradient of the mixture; a caller that differentiates with respect to
        # the input waveform gets the module-by-module path, which does
        wants_input_grad = torch.is_grad_enabled() and input.requires_grad
        if input.dim() == 3 and self._fused_ok() and not wants_input_grad:
            batch_size, C_in, T = input.size()
            assert C_in == 1, "input.size() is expected (?, 1, ?), but given {}".format(input.size())
            return ConvTasNetFn.apply(input, self.fused_config(), *self.fused_parameters())
        output, _ = self.extract_latent(input)
        return output

    # ------------------------------------------------------------------ modular path
    def extract_latent(self, input):
        """
        Args:
            input (batch_size, C_in, T)
        Returns:
            output (batch_size, n_sources, T) or (batch_size, n_sources, C_in, T)
            latent (batch_size, n_sources, n_basis, T'), where T' = (T-K)//S+1
        """
        rows, mics = waveform_rows(input)
        left, right = frame_padding(rows.size(-1), self.kernel_size, self.stride)
        w = self.encoder(F.pad(rows, (left, right)))
        if torch.is_complex(w):     # complex Fourier latent: the mask scales the magnitude, the phase is kept (:152-156)
            magnitude = torch.abs(w)
            latent = (magnitude.unsqueeze(1) * self.separator(magnitude)) * torch.exp(1j * torch.angle(w)).unsqueeze(1)
        else:
            latent = self.separator(w) * w.unsqueeze(1)
        return decode_sources(self.decoder, latent, mics, left, right), latent

    def get_config(self):
        config = {
            'in_channels': self.in_channels,
            'n_basis': self.n_basis,
            'kernel_size': self.kernel_size, 'stride': self.stride,
            'enc_basis': self.enc_basis, 'dec_basis': self.dec_basis,
            'enc_nonlinear': self.enc_nonlinear,
            'window_fn': self.window_fn,
            'enc_onesided': self.enc_onesided, 'enc_return_complex': self.enc_return_complex,
            'sep_hidden_channels': self.sep_hidden_channels,
            'sep_bottleneck_channels': self.sep_bottleneck_channels,
            'sep_skip_channels': self.sep_skip_channels,
            'sep_kernel_size': self.sep_kernel_size,
            'sep_num_blocks': self.sep_num_blocks, 'sep_num_layers': self.sep_num_layers,
            'dilated': self.dilated, 'separable': self.separable,
            'causal': self.causal,
            'sep_nonlinear': self.sep_nonlinear,
            'sep_norm': self.sep_norm,
            'mask_nonlinear': self.mask_nonlinear,
            'n_sources': self.n_sources,
            'eps': self.eps
        }
        return config

    def get_package(self):
        return self.get_config()

    @classmethod
    def build_model(cls, model_path, load_state_dict=False):
        config = torch.load(model_path, map_location=lambda storage, loc: storage)

        in_channels = config.get('in_channels') or 1
        n_basis = config.get('n_bases') or config['n_basis']
        kernel_size, stride = config['kernel_size'], config['stride']
        enc_basis = config.get('enc_bases') or config['enc_basis']
        dec_basis = config.get('dec_bases') or config['dec_basis']
        enc_nonlinear = config['enc_nonlinear']
        enc_onesided = config.get('enc_onesided') or None
        enc_return_complex = config.get('enc_return_complex') or None
        window_fn = config['window_fn']

        model = cls(
            n_basis, in_channels=in_channels, kernel_size=kernel_size, stride=stride, enc_basis=enc_basis,
            dec_basis=dec_basis, enc_nonlinear=enc_nonlinear, window_fn=window_fn, enc_onesided=enc_onesided,
            enc_return_complex=enc_return_complex,
            sep_hidden_channels=config['sep_hidden_channels'],
            sep_bottleneck_channels=config['sep_bottleneck_channels'],
            sep_skip_channels=config['sep_skip_channels'], sep_kernel_size=config['sep_kernel_size'],
            sep_num_blocks=config['sep_num_blocks'], sep_num_layers=config['sep_num_layers'],
            dilated=config['dilated'], separable=config['separable'], causal=config['causal'],
            sep_nonlinear=config['sep_nonlinear'], sep_norm=config['sep_norm'],
            mask_nonlinear=config['mask_nonlinear'], n_sources=config['n_sources'], eps=config['eps'])

        if load_state_dict:
            model.load_state_dict(config['state_dict'])
        return model

    @classmethod
    def build_from_pretrained(cls, root="./pretrained", quiet=False, load_state_dict=True, **kwargs):
        raise NotImplementedError("pretrained checkpoints are fetched from the network by the reference "
                                  "(src/models/conv_tasnet.py:238-310); load a local file with build_model()")

    @property
    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class Separator(nn.Module):
    """src/models/conv_tasnet.py:322-378"""

    def __init__(self, num_features, bottleneck_channels=128, hidden_channels=256, skip_channels=128, kernel_size=3,
                 num_blocks=3, num_layers=8, dilated=True, separable=True, causal=True, nonlinear='prelu', norm=True,
                 mask_nonlinear='sigmoid', n_sources=2, eps=EPS):
        super().__init__()
        self.num_features, self.n_sources = num_features, n_sources
        self.num_blocks, self.num_layers = num_blocks, num_layers
        self.causal, self.eps = causal, eps
        self.hip_ok = nonlinear == 'prelu' and norm
        self.dilated, self.separable = dilated, separable

        norm_name = 'cLN' if causal else 'gLN'
        self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps)
        self.bottleneck_conv1d = nn.Conv1d(num_features, bottleneck_channels, kernel_size=1, stride=1)
        self.tdcn = TimeDilatedConvNet(
            bottleneck_channels, hidden_channels=hidden_channels, skip_channels=skip_channels,
            kernel_size=kernel_size, num_blocks=num_blocks, num_layers=num_layers, dilated=dilated,
            separable=separable, causal=causal, nonlinear=nonlinear, norm=norm)
        self.prelu = nn.PReLU()
        self.mask_conv1d = nn.Conv1d(skip_channels, n_sources * num_features, kernel_size=1, stride=1)

        if mask_nonlinear == 'sigmoid':
            kwargs = {}
        elif mask_nonlinear == 'softmax':
            kwargs = {"dim": 1}
        else:
            raise ValueError("Cannot support {}".format(mask_nonlinear))
        self.mask_nonlinear = choose_nonlinear(mask_nonlinear, **kwargs)

    def layer_parameters(self):
        ps = []
        for block in self.tdcn.net:
            for layer in block.net:
                ps += layer.parameter_list()
        return ps

    def forward(self, input):
        """
        Args:
            input (batch_size, num_features, n_frames)
        Returns:
            output (batch_size, n_sources, n_basis, n_frames)
        """
        if not self.hip_ok:
            raise NotImplementedError("HIP path: Separator needs prelu and norm")
        num_features, n_sources = self.num_features, self.n_sources
        batch_size, _, n_frames = input.size()
        if not self.dilated or not self.separable:
            # stride-2 layers (src/models/tdcn.py:55-57) or dense dilated heads (:102-105): module by module,
            # conv_tasnet.py:360-376
            g0, be0 = _norm_params(self.norm1d)
            x = LayerNormFn.apply(input, g0, be0, self.eps, self.causal)
            x = Conv1x1Fn.apply(x, self.bottleneck_conv1d.weight, self.bottleneck_conv1d.bias)
            x = self.tdcn(x)
            x = F.prelu(x, self.prelu.weight)
            x = Conv1x1Fn.apply(x, self.mask_conv1d.weight, self.mask_conv1d.bias)
            x = self.mask_nonlinear(x)
            return x.reshape(batch_size, n_sources, num_features, n_frames)
        cfg = dict(sep_num_blocks=self.num_blocks, sep_num_layers=self.num_layers, causal=self.causal,
                   n_sources=n_sources, eps=self.eps)
        g0, be0 = _norm_params(self.norm1d)
        params = [g0, be0, self.bottleneck_conv1d.weight, self.bottleneck_conv1d.bias] + self.layer_parameters() + \
                 [self.prelu.weight, self.mask_conv1d.weight, self.mask_conv1d.bias]
        x = SeparatorFn.apply(input, cfg, *params)
        x = self.mask_nonlinear(x)    # over all n_sources*num_features channels for softmax (conv_tasnet.py:346,375)
        return x.reshape(batch_size, n_sources, num_features, n_frames)
