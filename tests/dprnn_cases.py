"""DPRNN-TasNet configurations shared by tools/gen_golden_dprnn.py and the tests."""

SMALL = dict(n_basis=16, kernel_size=4, stride=2, enc_basis='trainable', dec_basis='trainable', enc_nonlinear='relu',
             sep_hidden_channels=24, sep_bottleneck_channels=16, sep_chunk_size=10, sep_hop_size=5, sep_num_blocks=2,
             sep_norm=True, mask_nonlinear='sigmoid', causal=False, n_sources=2)
SMALL_CAUSAL3 = dict(SMALL, causal=True, n_sources=3, enc_nonlinear=None, sep_chunk_size=12, sep_hop_size=4)
# defaults of the reference recipe (egs/wsj0-mix/dprnn-tasnet/train.sh:28-39), 16 kHz
PAPER = dict(n_basis=64, kernel_size=2, stride=1, enc_basis='trainable', dec_basis='trainable', enc_nonlinear=None,
             sep_hidden_channels=128, sep_bottleneck_channels=64, sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6,
             sep_norm=True, mask_nonlinear='sigmoid', causal=False, n_sources=2)
E2E = {"small": (SMALL, 2, 403), "small_causal3": (SMALL_CAUSAL3, 2, 250), "paper": (PAPER, 1, 64000)}
# (batch, channels, frames, chunk, hop)
SEGMENT_CASES = {"seg_even": (2, 3, 40, 10, 5), "seg_ragged": (1, 4, 53, 12, 4), "seg_nooverlap": (2, 2, 30, 10, 10),
                 "seg_long": (1, 2, 4001, 250, 125)}
