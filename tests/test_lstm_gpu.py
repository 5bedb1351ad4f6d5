"""The persistent LSTM recurrence (ctn_lstm_fwd / ctn_lstm_bwd through speech_amd.models.dprnn.LSTM) against torch's own
nn.LSTM run in float64 on the CPU with the same parameters: outputs and every gradient, for hidden sizes that fill and
do not fill the kernels' padded tiles, sequence counts that do not fill a 16-sequence workgroup, one and two directions,
and the dual-path blocks' own shape (hidden 128)."""
import pytest
import torch
import torch.nn as nn

from conftest import assert_close
from speech_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N,T,F,H,bidir", [(5, 7, 16, 24, True), (37, 50, 16, 24, False), (16, 1, 8, 64, True),
                                           (33, 20, 64, 128, True), (3, 250, 64, 128, True), (100, 33, 12, 100, True),
                                           (17, 9, 64, 4, False)])
def test_lstm_matches_float64(N, T, F, H, bidir):
    from speech_amd.models.dprnn import LSTM
    torch.manual_seed(H + T)
    ref = nn.LSTM(F, H, batch_first=True, bidirectional=bidir).double()
    mod = LSTM(F, H, batch_first=True, bidirectional=bidir)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    mod.to(DEV)
    x = synth.normal((N, T, F), 7).double().requires_grad_(True)
    gy = synth.normal((N, T, (2 if bidir else 1) * H), 8).double()
    y64, (h64, c64) = ref(x)
    (y64 * gy).sum().backward()
    xg = x.detach().float().to(DEV).requires_grad_(True)
    assert mod._persistent_ok(xg, None)
    y, (h_n, c_n) = mod(xg)                    # the final states, as nn.LSTM returns them
    assert_close(h_n, h64, 2e-5, "h_n")
    assert_close(c_n, c64, 2e-5, "c_n")
    (y * gy.float().to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert_close(y, y64, 2e-5, "h")
    assert_close(xg.grad, x.grad, 5e-5, "dx")
    for (k, p), (_, p64) in zip(mod.named_parameters(), ref.named_parameters()):
        assert_close(p.grad, p64.grad, 5e-5, "d" + k)


def test_lstm_falls_back_to_the_library_for_what_the_kernels_do_not_cover():
    from speech_amd.models.dprnn import LSTM
    mod = LSTM(8, 160, batch_first=True, bidirectional=True).to(DEV)       # hidden size above the kernels' 128
    x = synth.normal((4, 6, 8), 3).to(DEV)
    assert not mod._persistent_ok(x, None)
    with pytest.warns(RuntimeWarning, match="library kernels"):
        y, (h, c) = mod(x)
    assert tuple(y.shape) == (4, 6, 320) and tuple(h.shape) == (2, 4, 160)
    # other dtypes and packed input are outside the kernels as well, and must not reach them
    small = LSTM(8, 16, batch_first=True).to(DEV)
    assert small._persistent_ok(x, None)
    assert not small._persistent_ok(x.double(), None)
    assert not small.double()._persistent_ok(x, None)
    packed = nn.utils.rnn.pack_padded_sequence(x.double(), torch.tensor([6, 5, 3, 1]), batch_first=True)
    assert not small._persistent_ok(packed, None)
