"""Dual-path RNN blocks (interface and state_dict keys of src/models/dprnn.py:6-148 of the reference).

A block applies the same three-stage unit twice, along two different axes of a (batch, features, chunks S, chunk size K)
tensor: a recurrence along ONE of the two time axes with the other folded into the batch, a linear map back to the
feature count, a layer norm over (features, S*K), and a residual connection.  `_PathRNN` is that unit; the intra-chunk
block walks K (bidirectional always), the inter-chunk block walks S (one direction only when causal, and then with the
cumulative norm).

The recurrence runs on this package's persistent LSTM kernels (`LSTM` below: nn.LSTM's parameters, ctn_lstm_fwd/bwd
through speech_amd.functional.LstmFn) -- one launch per sequence walk instead of two library launches per time step --
with the input projection of all steps and the weight gradients as single large GEMMs.  The norms are this package's
kernels as well, and so are the axis shuffles between the two layouts (tiled transposes, ctn_permute); the linear layer
is torch's (a plain library GEMM).
"""
import warnings

import torch
import torch.nn as nn

from .. import ops
from ..functional import FeatureFirstFn, LinearFn, LstmFn, RowsFromBlocksFn, SwapAddFn
from ..utils.tasnet import choose_layer_norm

EPS = 1e-12


class LSTM(nn.LSTM):
    """nn.LSTM as a parameter container -- same names, shapes and initialisation, so checkpoints interchange with the
    reference's -- with the forward pass on the persistent recurrence kernels.  What those kernels do not cover (an
    initial state, several layers, projections, hidden sizes above 128 or not a multiple of 4, time-major or packed
    input, other dtypes than fp32) takes nn.LSTM's own forward, i.e. the GPU library's LSTM, and says so once per
    module with a warning: that path is not this package's kernels."""

    _warned = False

    def _persistent_ok(self, input, hx):
        return (hx is None and isinstance(input, torch.Tensor) and input.is_cuda and input.dim() == 3
                and input.dtype == torch.float32 and all(w.dtype == torch.float32 for w in self._flat_weights)
                and self.batch_first and self.num_layers == 1
                and self.bias and not self.proj_size and self.hidden_size <= ops.LSTM_MAX_HIDDEN
                and self.hidden_size % 4 == 0 and (not self.training or self.dropout == 0.0))

    def forward(self, input, hx=None):
        if not self._persistent_ok(input, hx):
            if not self._warned:
                self._warned = True
                warnings.warn("speech_amd LSTM: this call is outside what the persistent recurrence kernels cover (one "
                              "layer, fp32, batch_first 3-D tensor input, hidden size <= {} and a multiple of 4, no initial "
                              "state, no projection); it runs on torch.nn.LSTM's library kernels instead"
                              .format(ops.LSTM_MAX_HIDDEN), RuntimeWarning, stacklevel=2)
            return super().forward(input, hx)
        n_dir = 2 if self.bidirectional else 1
        output, h_n, c_n = LstmFn.apply(input, n_dir, *self._flat_weights)
        return output, (h_n, c_n)

    def flatten_parameters(self):
        """The library LSTM wants its weights in one buffer; the persistent kernels read them where they are."""
        if not all(w.is_cuda for w in self._flat_weights) or self.hidden_size > ops.LSTM_MAX_HIDDEN or self.hidden_size % 4:
            super().flatten_parameters()


class _PathRNN(nn.Module):
    """Recurrence along one axis of (batch, features, S, K), linear map, norm, residual.

    walk_chunks=False: sequences are the chunks' interiors (length K, batch * S of them): src/models/dprnn.py:76-94.
    walk_chunks=True: sequences run across chunks (length S, batch * K of them): src/models/dprnn.py:128-148; the norm
    then sees the frames in (K, S) order, as the reference lays them out before normalising."""

    def __init__(self, num_features, hidden_channels, walk_chunks, bidirectional, norm, norm_name, causal, rnn_type, eps):
        super().__init__()
        if rnn_type != 'lstm':
            raise NotImplementedError("Not support {}.".format(rnn_type))
        self.num_features, self.hidden_channels = num_features, hidden_channels
        self.walk_chunks = walk_chunks
        self.norm = norm
        self.rnn = LSTM(input_size=num_features, hidden_size=hidden_channels, batch_first=True, bidirectional=bidirectional)
        self.fc = nn.Linear((2 if bidirectional else 1) * hidden_channels, num_features)
        if norm:
            self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps)

    def forward(self, input):
        batch_size, num_features, S, K = input.size()
        outer, steps = (K, S) if self.walk_chunks else (S, K)
        sequences = RowsFromBlocksFn.apply(input, self.walk_chunks)        # (batch * outer, steps, features)
        y, _ = self.rnn(sequences)
        y = LinearFn.apply(y, self.fc.weight, self.fc.bias)                # (sequences, steps, features)
        y = FeatureFirstFn.apply(y.view(batch_size, outer * steps, num_features))      # (batch, features, outer * steps)
        if self.norm:
            y = self.norm1d(y)
        y = y.view(batch_size, num_features, outer, steps)
        if self.walk_chunks:
            return SwapAddFn.apply(y, input)                               # back to (batch, features, S, K), plus input
        return y + input


class IntraChunkRNN(_PathRNN):
    """src/models/dprnn.py:50-95"""

    def __init__(self, num_features, hidden_channels, norm=True, rnn_type='lstm', eps=EPS):
        super().__init__(num_features, hidden_channels, walk_chunks=False, bidirectional=True, norm=norm, norm_name='gLN',
                         causal=False, rnn_type=rnn_type, eps=eps)


class InterChunkRNN(_PathRNN):
    """src/models/dprnn.py:97-148"""

    def __init__(self, num_features, hidden_channels, causal, norm=True, rnn_type='lstm', eps=EPS):
        super().__init__(num_features, hidden_channels, walk_chunks=True, bidirectional=not causal, norm=norm,
                         norm_name='cLN' if causal else 'gLN', causal=causal, rnn_type=rnn_type, eps=eps)


class DPRNNBlock(nn.Module):
    """src/models/dprnn.py:30-48: the intra-chunk unit, then the inter-chunk unit."""

    def __init__(self, num_features, hidden_channels, causal, norm=True, rnn_type='lstm', eps=EPS):
        super().__init__()
        self.intra_chunk_block = IntraChunkRNN(num_features, hidden_channels, norm=norm, rnn_type=rnn_type, eps=eps)
        self.inter_chunk_block = InterChunkRNN(num_features, hidden_channels, causal=causal, norm=norm, rnn_type=rnn_type,
                                               eps=eps)

    def forward(self, input):
        return self.inter_chunk_block(self.intra_chunk_block(input))


class DPRNN(nn.Module):
    """src/models/dprnn.py:8-28: num_blocks dual-path blocks in sequence, (batch, features, S, K) in and out."""

    def __init__(self, num_features, hidden_channels, num_blocks=6, norm=True, causal=False, rnn_type='lstm', eps=EPS):
        super().__init__()
        self.net = nn.Sequential(*[DPRNNBlock(num_features, hidden_channels, causal=causal, norm=norm, rnn_type=rnn_type,
                                              eps=eps) for _ in range(num_blocks)])

    def forward(self, input):
        return self.net(input)
