"""Dual-path RNN blocks with the reference's interface and state_dict keys (src/models/dprnn.py:6-148).

The layer norms run on this package's HIP kernels (on the (B, C, S*K) view the reference normalises); the recurrences
are torch's nn.LSTM, i.e. MIOpen on the GPU, and the linear layers torch's nn.Linear (hipBLASLt): plain library
calls, with no hand-written kernel here yet.  Permutations and the residual additions are torch tensor operations.
"""
import torch.nn as nn

from ..utils.model import choose_rnn
from ..utils.tasnet import choose_layer_norm

EPS = 1e-12


class DPRNN(nn.Module):
    """src/models/dprnn.py:8-28"""

    def __init__(self, num_features, hidden_channels, num_blocks=6, norm=True, causal=False, rnn_type='lstm', eps=EPS):
        super().__init__()
        net = []
        for _ in range(num_blocks):
            net.append(DPRNNBlock(num_features, hidden_channels, norm=norm, causal=causal, rnn_type=rnn_type, eps=eps))
        self.net = nn.Sequential(*net)

    def forward(self, input):
        """(batch_size, num_features, S, chunk_size) -> the same shape"""
        return self.net(input)


class DPRNNBlock(nn.Module):
    """src/models/dprnn.py:30-48"""

    def __init__(self, num_features, hidden_channels, causal, norm=True, rnn_type='lstm', eps=EPS):
        super().__init__()
        self.intra_chunk_block = IntraChunkRNN(num_features, hidden_channels, norm=norm, rnn_type=rnn_type, eps=eps)
        self.inter_chunk_block = InterChunkRNN(num_features, hidden_channels, norm=norm, causal=causal, rnn_type=rnn_type,
                                               eps=eps)

    def forward(self, input):
        return self.inter_chunk_block(self.intra_chunk_block(input))


class IntraChunkRNN(nn.Module):
    """src/models/dprnn.py:50-95: bidirectional recurrence along each chunk, linear back to num_features, gLN over
    (C, S*K), residual."""

    def __init__(self, num_features, hidden_channels, norm=True, rnn_type='lstm', eps=EPS):
        super().__init__()
        self.num_features, self.hidden_channels = num_features, hidden_channels
        num_directions = 2
        self.norm = norm
        if rnn_type == 'lstm':
            self.rnn = choose_rnn(rnn_type, input_size=num_features, hidden_size=hidden_channels, batch_first=True,
                                  bidirectional=True)
        else:
            raise NotImplementedError("Not support {}.".format(rnn_type))
        self.fc = nn.Linear(num_directions * hidden_channels, num_features)
        if self.norm:
            self.norm1d = choose_layer_norm('gLN', num_features, causal=False, eps=eps)

    def forward(self, input):
        num_features = self.num_features
        batch_size, _, S, chunk_size = input.size()
        self.rnn.flatten_parameters()
        residual = input
        x = input.permute(0, 2, 3, 1).contiguous()                 # (batch_size, S, chunk_size, num_features)
        x = x.view(batch_size * S, chunk_size, num_features)
        x, _ = self.rnn(x)                                         # (batch_size*S, chunk_size, 2*hidden_channels)
        x = self.fc(x)
        x = x.view(batch_size, S * chunk_size, num_features)
        x = x.permute(0, 2, 1).contiguous()                        # (batch_size, num_features, S*chunk_size)
        if self.norm:
            x = self.norm1d(x)
        x = x.reshape(batch_size, num_features, S, chunk_size)
        return x + residual


class InterChunkRNN(nn.Module):
    """src/models/dprnn.py:97-148: recurrence across chunks (one direction if causal), linear, gLN / cLN over
    (C, K*S) in chunk-major order as the reference lays it out, residual."""

    def __init__(self, num_features, hidden_channels, causal, norm=True, rnn_type='lstm', eps=EPS):
        super().__init__()
        self.num_features, self.hidden_channels = num_features, hidden_channels
        self.norm = norm
        if rnn_type != 'lstm':
            raise NotImplementedError("Not support {}.".format(rnn_type))
        num_directions = 1 if causal else 2
        self.rnn = choose_rnn(rnn_type, input_size=num_features, hidden_size=hidden_channels, batch_first=True,
                              bidirectional=not causal)
        self.fc = nn.Linear(num_directions * hidden_channels, num_features)
        if self.norm:
            norm_name = 'cLN' if causal else 'gLN'
            self.norm1d = choose_layer_norm(norm_name, num_features, causal=causal, eps=eps)

    def forward(self, input):
        num_features = self.num_features
        batch_size, _, S, chunk_size = input.size()
        self.rnn.flatten_parameters()
        residual = input
        x = input.permute(0, 3, 2, 1).contiguous()                 # (batch_size, chunk_size, S, num_features)
        x = x.view(batch_size * chunk_size, S, num_features)
        x, _ = self.rnn(x)
        x = self.fc(x)
        x = x.view(batch_size, chunk_size * S, num_features)
        x = x.permute(0, 2, 1).contiguous()                        # (batch_size, num_features, chunk_size*S)
        if self.norm:
            x = self.norm1d(x)
        x = x.reshape(batch_size, num_features, chunk_size, S)
        x = x.permute(0, 1, 3, 2).contiguous()
        return x + residual
