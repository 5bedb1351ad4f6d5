"""Filterbank factory with the reference's interface (src/utils/filterbank.py:5-67)."""
from ..models.filterbank import FourierEncoder, FourierDecoder, Encoder, Decoder, GatedEncoder, PinvDecoder

EPS = 1e-12
_FOURIER = ['Fourier', 'trainableFourier', 'trainableFourierTrainablePhase']


def _fourier_options(basis, hidden_channels, kwargs):
    onesided, return_complex = bool(kwargs['enc_onesided']), bool(kwargs['enc_return_complex'])
    return dict(n_basis=compute_valid_basis(hidden_channels, onesided=onesided, return_complex=return_complex),
                window_fn=kwargs['window_fn'], trainable=basis != 'Fourier',
                trainable_phase=basis == 'trainableFourierTrainablePhase', onesided=onesided), return_complex


def choose_filterbank(hidden_channels, kernel_size, stride=None, enc_basis='trainable', dec_basis='trainable', **kwargs):
    in_channels = kwargs.get('in_channels') or 1

    if enc_basis == 'trainable':
        nonlinear = None if dec_basis == 'pinv' else kwargs['enc_nonlinear']
        encoder = Encoder(in_channels, hidden_channels, kernel_size, stride=stride, nonlinear=nonlinear)
    elif enc_basis in _FOURIER:
        assert_monoral(in_channels)
        opts, return_complex = _fourier_options(enc_basis, hidden_channels, kwargs)
        n_basis = opts.pop('n_basis')
        encoder = FourierEncoder(n_basis, kernel_size, stride=stride, return_complex=return_complex, **opts)
    elif enc_basis == 'trainableGated':
        encoder = GatedEncoder(in_channels, hidden_channels, kernel_size=kernel_size, stride=stride,
                               eps=kwargs.get('eps') or EPS)
    else:
        raise NotImplementedError("Not support {} for encoder".format(enc_basis))

    if dec_basis == 'trainable':
        decoder = Decoder(hidden_channels, in_channels, kernel_size, stride=stride)
    elif dec_basis in _FOURIER:
        assert_monoral(in_channels)
        opts, _ = _fourier_options(dec_basis, hidden_channels, kwargs)
        n_basis = opts.pop('n_basis')
        decoder = FourierDecoder(n_basis, kernel_size, stride=stride, **opts)
    elif dec_basis == 'pinv':
        if enc_basis in ['trainable', 'trainableFourier', 'trainableFourierTrainablePhase']:
            assert_monoral(in_channels)
            decoder = PinvDecoder(encoder)
        else:
            raise NotImplementedError("Not support {} for decoder".format(dec_basis))
    else:
        raise NotImplementedError("Not support {} for decoder".format(dec_basis))

    return encoder, decoder


def assert_monoral(in_channels):
    assert in_channels == 1, "`in_channels` is expected 1, but given {}.".format(in_channels)


def compute_valid_basis(hidden_channels, onesided=True, return_complex=True):
    """The transform length whose output has `hidden_channels` channels (src/utils/filterbank.py:52-67)."""
    if not return_complex:
        assert hidden_channels % 2 == 0, "`hidden_channels` is expected even."
    if onesided:
        if return_complex:
            assert hidden_channels % 2 == 1, "`hidden_channels` is expected odd."
            return 2 * (hidden_channels - 1)
        return 2 * (hidden_channels // 2 - 1)
    return hidden_channels if return_complex else hidden_channels // 2
