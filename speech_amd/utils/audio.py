"""Analysis windows for the Fourier filterbanks (src/utils/audio.py:9-43)."""
import torch


def build_window(n_fft, window_fn='hann', **kwargs):
    if window_fn in ('hann', 'hamming'):
        assert set(kwargs) == set(), "kwargs is expected empty but given kwargs={}.".format(kwargs)
        fn = torch.hann_window if window_fn == 'hann' else torch.hamming_window
        return fn(n_fft, periodic=True)
    if window_fn == 'blackman':
        return torch.blackman_window(n_fft, periodic=True)
    if window_fn == 'kaiser':
        assert set(kwargs) == {'beta'}, "kwargs is expected to include key `beta` but given kwargs={}.".format(kwargs)
        return torch.kaiser_window(n_fft, beta=kwargs['beta'], periodic=True)
    raise ValueError("Not support {} window.".format(window_fn))


def build_optimal_window(window, hop_length=None):
    """The synthesis window that makes overlap-add with `window` an identity: window / sum over the hops of window^2.
    Args:
        window: (window_length,)
    """
    window_length = len(window)
    if hop_length is None:
        hop_length = window_length // 2
    energy = torch.zeros_like(window)
    for idx in range(window_length // hop_length):
        energy = energy + torch.roll(window, hop_length * idx) ** 2
    return window / energy
