"""Segment1d / OverlapAdd1d of the reference (src/models/transform.py:6-61) on the HIP chunking kernels."""
import torch.nn as nn

from ..functional import SegmentFn, OverlapAddFn


class Segment1d(nn.Module):
    """Segmentation. Input tensor is 3-D (audio-like), but output tensor is 4-D (image-like)."""

    def __init__(self, chunk_size, hop_size):
        super().__init__()
        self.chunk_size, self.hop_size = chunk_size, hop_size

    def forward(self, input):
        """
        Args:
            input (batch_size, num_features, n_frames)
        Returns:
            output (batch_size, num_features, S, chunk_size), S = (n_frames-chunk_size)//hop_size + 1
        """
        n_frames = input.size(2)
        S = (n_frames - self.chunk_size) // self.hop_size + 1
        return SegmentFn.apply(input, self.chunk_size, self.hop_size, S, 0)

    def extra_repr(self):
        return "chunk_size={chunk_size}, hop_size={hop_size}".format(chunk_size=self.chunk_size, hop_size=self.hop_size)


class OverlapAdd1d(nn.Module):
    """Overlap-add operation. Input tensor is 4-D (image-like), but output tensor is 3-D (audio-like)."""

    def __init__(self, chunk_size, hop_size):
        super().__init__()
        self.chunk_size, self.hop_size = chunk_size, hop_size

    def forward(self, input):
        """
        Args:
            input: (batch_size, num_features, S, chunk_size)
        Returns:
            output: (batch_size, num_features, n_frames), n_frames = (S-1)*hop_size + chunk_size
        """
        batch_size, num_features, S, chunk_size = input.size()
        n_frames = (S - 1) * self.hop_size + chunk_size
        return OverlapAddFn.apply(input, self.hop_size, n_frames, 0)

    def extra_repr(self):
        return "chunk_size={chunk_size}, hop_size={hop_size}".format(chunk_size=self.chunk_size, hop_size=self.hop_size)
