"""DPRNN-TasNet with the reference's interface and state_dict keys (src/models/dprnn_tasnet.py:15-350).

Encoder, decoder, the separator's entry norm, its two pointwise convolutions and the chunking / overlap-add with their
paddings run on this package's HIP kernels, shared with Conv-TasNet; the dual-path blocks in between are described in
models/dprnn.py; the PReLU, the mask non-linearity and the mask multiplication are torch tensor operations.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functional import Conv1x1Fn, SegmentFn, OverlapAddFn
from ..utils.filterbank import choose_filterbank
from ..utils.tasnet import choose_layer_norm, frame_padding, waveform_rows, decode_sources
from .dprnn import DPRNN
from .transform import Segment1d, OverlapAdd1d

EPS = 1e-12


class DPRNNTasNet(nn.Module):
    def __init__(self, n_basis, kernel_size, stride=None, enc_basis=None, dec_basis=None, sep_hidden_channels=128,
                 sep_bottleneck_channels=64, sep_chunk_size=100, sep_hop_size=50, sep_num_blocks=6, sep_norm=True,
                 mask_nonlinear='sigmoid', causal=True, rnn_type='lstm', n_sources=2, eps=EPS, **kwargs):
        super().__init__()
        if stride is None:
            stride = kernel_size // 2
        assert kernel_size % stride == 0, "kernel_size is expected divisible by stride"

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

        self.sep_hidden_channels, self.sep_bottleneck_channels = sep_hidden_channels, sep_bottleneck_channels
        self.sep_chunk_size, self.sep_hop_size = sep_chunk_size, sep_hop_size
        self.sep_num_blocks = sep_num_blocks
        self.causal = causal
        self.sep_norm = sep_norm
        self.mask_nonlinear = mask_nonlinear
        self.rnn_type = rnn_type
        self.n_sources = n_sources
        self.eps = eps

        encoder, decoder = choose_filterbank(n_basis, kernel_size=kernel_size, stride=stride, enc_basis=enc_basis,
                                             dec_basis=dec_basis, **kwargs)
        self.encoder = encoder
        self.separator = Separator(
            n_basis, bottleneck_channels=sep_bottleneck_channels, hidden_channels=sep_hidden_channels,
            chunk_size=sep_chunk_size, hop_size=sep_hop_size, num_blocks=sep_num_blocks, norm=sep_norm,
            mask_nonlinear=mask_nonlinear, causal=causal, rnn_type=rnn_type, n_sources=n_sources, eps=eps)
        self.decoder = decoder

    def forward(self, input):
        output, _ = self.extract_latent(input)
        return output

    def extract_latent(self, input):
        """
        Args:
            input (batch_size, 1, T)
        Returns:
            output (batch_size, n_sources, T)
            latent (batch_size, n_sources, n_basis, T'), where T' = (T-K)//S+1
        """
        rows, mics = waveform_rows(input)
        left, right = frame_padding(rows.size(-1), self.kernel_size, self.stride)
        w = self.encoder(F.pad(rows, (left, right)))
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
            'sep_chunk_size': self.sep_chunk_size, 'sep_hop_size': self.sep_hop_size,
            'sep_num_blocks': self.sep_num_blocks,
            'causal': self.causal,
            'sep_norm': self.sep_norm,
            'mask_nonlinear': self.mask_nonlinear,
            'rnn_type': self.rnn_type,
            'n_sources': self.n_sources,
            'eps': self.eps
        }
        return config

    @classmethod
    def build_model(cls, model_path, load_state_dict=False):
        config = torch.load(model_path, map_location=lambda storage, loc: storage)
        in_channels = config.get('in_channels') or 1
        n_basis = config.get('n_bases') or config['n_basis']
        kernel_size, stride = config['kernel_size'], config['stride']
        enc_basis = config.get('enc_bases') or config['enc_basis']
        dec_basis = config.get('dec_bases') or config['dec_basis']
        model = cls(
            n_basis, in_channels=in_channels, kernel_size=kernel_size, stride=stride, enc_basis=enc_basis,
            dec_basis=dec_basis, enc_nonlinear=config['enc_nonlinear'], window_fn=config['window_fn'],
            enc_onesided=config.get('enc_onesided') or None, enc_return_complex=config.get('enc_return_complex') or None,
            sep_hidden_channels=config['sep_hidden_channels'], sep_bottleneck_channels=config['sep_bottleneck_channels'],
            sep_chunk_size=config['sep_chunk_size'], sep_hop_size=config['sep_hop_size'],
            sep_num_blocks=config['sep_num_blocks'], sep_norm=config['sep_norm'], mask_nonlinear=config['mask_nonlinear'],
            causal=config['causal'], rnn_type=config.get('rnn_type', 'lstm'), n_sources=config['n_sources'],
            eps=config.get('eps') or EPS)
        if load_state_dict:
            model.load_state_dict(config['state_dict'])
        return model

    @classmethod
    def build_from_pretrained(cls, root="./pretrained", quiet=False, load_state_dict=True, **kwargs):
        raise NotImplementedError("pretrained checkpoints are fetched from the network by the reference "
                                  "(src/models/dprnn_tasnet.py:222-276); load a local file with build_model()")

    @property
    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class Separator(nn.Module):
    """src/models/dprnn_tasnet.py:288-350"""

    def __init__(self, num_features, bottleneck_channels=64, hidden_channels=128, chunk_size=100, hop_size=50,
                 num_blocks=6, norm=True, mask_nonlinear='sigmoid', causal=True, rnn_type='lstm', n_sources=2, eps=EPS):
        super().__init__()
        self.num_features, self.n_sources = num_features, n_sources
        self.chunk_size, self.hop_size = chunk_size, hop_size
        self.norm = norm

        norm_name = 'cLN' if causal else 'gLN'
        self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps)
        self.bottleneck_conv1d = nn.Conv1d(num_features, bottleneck_channels, kernel_size=1, stride=1)
        self.segment1d = Segment1d(chunk_size, hop_size)
        self.dprnn = DPRNN(bottleneck_channels, hidden_channels, num_blocks=num_blocks, causal=causal, norm=norm,
                           rnn_type=rnn_type, eps=eps)
        self.overlap_add1d = OverlapAdd1d(chunk_size, hop_size)
        self.prelu = nn.PReLU()
        self.mask_conv1d = nn.Conv1d(bottleneck_channels, n_sources * num_features, kernel_size=1, stride=1)
        if mask_nonlinear == 'sigmoid':
            self.mask_nonlinear = nn.Sigmoid()
        elif mask_nonlinear == 'softmax':
            self.mask_nonlinear = nn.Softmax(dim=1)
        else:
            raise ValueError("Cannot support {}".format(mask_nonlinear))

    def forward(self, input):
        """
        Args:
            input (batch_size, num_features, n_frames)
        Returns:
            output (batch_size, n_sources, num_features, n_frames)
        """
        num_features, n_sources = self.num_features, self.n_sources
        chunk_size, hop_size = self.chunk_size, self.hop_size
        batch_size, num_features, n_frames = input.size()

        padding = (hop_size - (n_frames - chunk_size) % hop_size) % hop_size
        padding_left = padding // 2
        S = (n_frames + padding - chunk_size) // hop_size + 1

        x = self.norm1d(input)
        x = Conv1x1Fn.apply(x, self.bottleneck_conv1d.weight, self.bottleneck_conv1d.bias)
        x = SegmentFn.apply(x, chunk_size, hop_size, S, padding_left)      # F.pad + Segment1d (:339-341)
        x = self.dprnn(x)
        x = OverlapAddFn.apply(x, hop_size, n_frames, padding_left)        # OverlapAdd1d + trimming F.pad (:343-344)
        x = self.prelu(x)
        x = Conv1x1Fn.apply(x, self.mask_conv1d.weight, self.mask_conv1d.bias)
        x = self.mask_nonlinear(x)
        return x.reshape(batch_size, n_sources, num_features, n_frames)
