"""Host logic of the epoch loop (speech_amd/driver.py) against decisions recorded from the reference's own trainer
(tests/golden/trainer.npz, written by tools/gen_golden_trainer.py): learning-rate halving, early stop, which epochs
write best.pth / last.pth; the Adam state's round trip through the torch.optim.Adam state_dict layout; and that the
checkpoint the reference wrote loads with the safe loader and names exactly this package's parameters."""
import os

import numpy as np
import pytest
import torch

import convtasnet_oracle as O
from test_host_logic import build

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def simulate(script, early_stop):
    """The loop of TrainerBase.run around EpochControl, without a model."""
    from speech_amd.driver import EpochControl
    c = EpochControl(early_stop=early_stop)
    lr, log, saved_best, last_epoch = 1e-3, [], [], 0
    for epoch, v in enumerate(script):
        if epoch > 0:
            log.append((lr, c.no_improvement, c.best_loss))
        save_best, halve, stop = c.update(v)
        if save_best:
            saved_best.append(epoch)
        if stop:
            break
        if halve:
            lr = 0.5 * lr
        last_epoch = epoch + 1
    log.append((lr, c.no_improvement, c.best_loss))
    return log, saved_best, last_epoch


@pytest.mark.parametrize("tag,early_stop", [("base", True), ("adhoc", False)])
@pytest.mark.parametrize("name", ["plateau", "long_plateau", "zigzag"])
def test_epoch_decisions_match_reference(golden, tag, early_stop, name):
    g = golden("trainer")
    script = g["script_" + name].tolist()
    log, saved_best, last_epoch = simulate(script, early_stop)
    key = "{}_{}_".format(tag, name)
    assert [e[0] for e in log] == g[key + "lr"].tolist()
    assert [e[1] for e in log] == g[key + "no_improvement"].tolist()
    assert [e[2] for e in log] == g[key + "best"].tolist()
    assert saved_best == g[key + "saved_best"].tolist()
    assert last_epoch == int(g[key + "last_epoch"])


def test_early_stop_happens_only_in_base_trainer(golden):
    g = golden("trainer")
    n = len(g["script_long_plateau"])
    assert int(g["base_long_plateau_last_epoch"]) < n          # the reference's TrainerBase stopped early ...
    assert int(g["adhoc_long_plateau_last_epoch"]) == n        # ... its AdhocTrainer did not


class FakeFlat:
    """The part of FlatTrainer that the state conversion touches."""

    def __init__(self, params):
        self.params = params
        self.m = {id(p): torch.randn_like(p) for p in params}
        self.v = {id(p): torch.rand_like(p) for p in params}
        self.step_count, self.betas, self.eps = 7, (0.9, 0.999), 1e-8

    def moment_views(self):
        return {id(p): (self.m[id(p)], self.v[id(p)]) for p in self.params}


def test_adam_state_round_trip_through_torch_layout():
    from speech_amd.driver import export_adam_state, import_adam_state
    model = build(O.CONFIG_SMALL)
    params = list(model.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    flat = FakeFlat(params)
    sd = export_adam_state(opt, flat)
    # a stock optimizer accepts it, and holds the same moments afterwards
    opt2 = torch.optim.Adam(params, lr=5e-4)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == 1e-3
    for p in params:
        assert torch.equal(opt2.state[p]["exp_avg"], flat.m[id(p)])
        assert torch.equal(opt2.state[p]["exp_avg_sq"], flat.v[id(p)])
        assert float(opt2.state[p]["step"]) == 7.0
    # and back
    flat2 = FakeFlat(params)
    flat2.step_count = 0
    opt3 = torch.optim.Adam(params, lr=1.0)
    import_adam_state(opt3, flat2, opt2.state_dict())
    assert flat2.step_count == 7 and opt3.param_groups[0]["lr"] == 1e-3
    for p in params:
        assert torch.equal(flat2.m[id(p)], flat.m[id(p)]) and torch.equal(flat2.v[id(p)], flat.v[id(p)])


def test_non_adam_optimizers_are_refused():
    from speech_amd.driver import _adam_group
    model = build(O.CONFIG_SMALL)
    with pytest.raises(NotImplementedError):
        _adam_group(torch.optim.SGD(model.parameters(), lr=0.1))
    with pytest.raises(NotImplementedError):
        _adam_group(torch.optim.Adam(model.parameters(), lr=0.1, weight_decay=1e-2))
    with pytest.raises(NotImplementedError):
        _adam_group(torch.optim.AdamW(model.parameters(), lr=0.1))


def test_reference_checkpoint_loads_safely_and_names_our_parameters(golden):
    g = golden("trainer")
    ckpt = torch.load(os.path.join(GOLDEN, "trainer_ref_last.pth"), map_location="cpu", weights_only=True)
    assert sorted(ckpt.keys()) == g["mini_ckpt_keys"].tolist()
    model = build(O.CONFIG_SMALL)
    assert list(ckpt["state_dict"].keys()) == list(model.state_dict().keys())
    model.load_state_dict(ckpt["state_dict"])
    assert ckpt["epoch"] == 2 and len(ckpt["optim_dict"]["state"]) == len(list(model.parameters()))
    for key in ("n_basis", "kernel_size", "stride", "sep_hidden_channels", "causal", "n_sources", "eps"):
        assert ckpt[key] == model.get_config()[key], key
