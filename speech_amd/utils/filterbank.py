"""Filterbank factory with the reference's interface (src/utils/filterbank.py:5-46); only the trainable
encoder / trainable decoder pair is implemented on this path."""
from ..models.filterbank import Encoder, Decoder

EPS = 1e-12


def choose_filterbank(hidden_channels, kernel_size, stride=None, enc_basis='trainable', dec_basis='trainable', **kwargs):
    in_channels = kwargs.get('in_channels') or 1
    if enc_basis == 'trainable':
        if dec_basis == 'pinv':
            raise NotImplementedError("pinv decoder is outside the HIP path")
        encoder = Encoder(in_channels, hidden_channels, kernel_size, stride=stride, nonlinear=kwargs['enc_nonlinear'])
    else:
        raise NotImplementedError("Not support {} for encoder".format(enc_basis))
    if dec_basis == 'trainable':
        decoder = Decoder(hidden_channels, in_channels, kernel_size, stride=stride)
    else:
        raise NotImplementedError("Not support {} for decoder".format(dec_basis))
    return encoder, decoder
