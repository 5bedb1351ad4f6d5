"""String -> torch.nn module lookups behind the model constructors (same names, arguments and failures as
src/utils/model.py:3-32 of the reference)."""
import torch.nn as nn

_ACTIVATIONS = {
    'relu': nn.ReLU,
    'sigmoid': nn.Sigmoid,
    'softmax': nn.Softmax,
    'tanh': nn.Tanh,
    'leaky-relu': nn.LeakyReLU,
    'gelu': nn.GELU,
}
_RECURRENT = {'rnn': nn.RNN, 'lstm': nn.LSTM, 'gru': nn.GRU}


def choose_nonlinear(name, **kwargs):
    """Activation module for `name`.  Only softmax takes arguments (its `dim`, which must be given); for every other
    name keyword arguments are ignored, as in the reference."""
    factory = _ACTIVATIONS.get(name)
    if factory is None:
        raise NotImplementedError("Invalid nonlinear function is specified. Choose 'relu' instead of {}.".format(name))
    if factory is nn.Softmax:
        assert 'dim' in kwargs, "dim is expected for softmax."
        return factory(**kwargs)
    return factory()


def choose_rnn(name, **kwargs):
    """nn.RNN / nn.LSTM / nn.GRU built with `kwargs`."""
    factory = _RECURRENT.get(name)
    if factory is None:
        raise NotImplementedError("Invalid RNN is specified. Choose 'rnn', 'lstm', or 'gru' instead of {}.".format(name))
    return factory(**kwargs)
