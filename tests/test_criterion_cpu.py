"""The CPU restatement of the criterion variants (oracle/criterion_oracle.py) against the reference's own outputs
(tests/golden/criterion.npz): losses within 1e-5, chosen indices bit-exact, gradients within 1e-4."""
import pytest
import torch

import criterion_oracle as CO
from conftest import assert_close
from criterion_cases import SDR_CASES, PIT_CASES, ORPIT_CASES, SINK_CASES
import convtasnet_oracle as O


def make(cname, reduction="mean"):
    fn = CO.sdr if cname.endswith("SDR") and "SI" not in cname else CO.sisdr
    return CO.Criterion(fn, negate=cname.startswith("Neg"), reduction=reduction)


def t(a):
    return torch.from_numpy(a)


@pytest.mark.parametrize("cname,red,bm", SDR_CASES)
def test_sdr_criteria(golden, cname, red, bm):
    g = golden("criterion")
    e = t(g["sdr_est"]).clone().requires_grad_(True)
    loss = make(cname, red)(e, t(g["sdr_tgt"]), batch_mean=bm)
    loss.sum().backward()
    key = "sdr_{}_{}_{}".format(cname, red, int(bm))
    assert loss.shape == g[key + "_loss"].shape
    assert_close(loss, g[key + "_loss"], 1e-5, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")


def test_sdr_input_ranks(golden):
    g = golden("criterion")
    loss = make("NegSDR")(t(g["sdr_est"])[:, 0], t(g["sdr_tgt"])[:, 0], batch_mean=False)
    assert_close(loss, g["sdr_rank2_loss"], 1e-5, "2-D input")
    loss = make("NegSDR")(t(g["sdr4_est"]), t(g["sdr4_tgt"]), batch_mean=False)
    assert loss.shape == g["sdr4_loss"].shape
    assert_close(loss, g["sdr4_loss"], 1e-5, "4-D input")


@pytest.mark.parametrize("S,cname,red", PIT_CASES)
def test_pit_over_other_criteria(golden, S, cname, red):
    g = golden("criterion")
    crit = make(cname, red)
    e = t(g["pit{}_est".format(S)]).clone().requires_grad_(True)
    tgt = t(g["pit{}_tgt".format(S)])
    key = "pit{}_{}_{}".format(S, cname, red)
    loss_b, perm = O.pit(crit, e, tgt, batch_mean=False, maximize=crit.maximize)
    loss, _ = O.pit(crit, e, tgt, maximize=crit.maximize)
    loss.backward()
    assert torch.equal(perm, t(g[key + "_perm"]))
    assert_close(loss_b, g[key + "_loss_b"], 1e-5, key)
    assert_close(loss, g[key + "_loss"], 1e-5, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")


@pytest.mark.parametrize("cname", ORPIT_CASES)
def test_orpit(golden, cname):
    g = golden("criterion")
    e = t(g["orpit_est"]).clone().requires_grad_(True)
    tgt = t(g["orpit_tgt"])
    key = "orpit_" + cname
    loss_b, idx = CO.orpit(make(cname), e, list(tgt), batch_mean=False)
    loss, _ = CO.orpit(make(cname), e, list(tgt))
    loss.backward()
    assert torch.equal(idx, t(g[key + "_idx"]))
    assert_close(loss_b, g[key + "_loss_b"], 1e-5, key)
    assert_close(loss, g[key + "_loss"], 1e-5, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")


def test_orpit_ragged_targets(golden):
    g = golden("criterion")
    tgt = t(g["orpit_tgt"])
    lens = [int(n) for n in g["orpit_packed_lens"]]
    e = t(g["orpit_packed_est"]).clone().requires_grad_(True)
    loss_b, idx = CO.orpit(make("NegSISDR"), e, [tgt[b, :n] for b, n in enumerate(lens)], batch_mean=False)
    loss_b.mean().backward()
    assert torch.equal(idx, t(g["orpit_packed_idx"]))
    assert_close(loss_b, g["orpit_packed_loss_b"], 1e-5, "ragged ORPIT")
    assert_close(e.grad, g["orpit_packed_dest"], 1e-4, "ragged ORPIT grad")


@pytest.mark.parametrize("S,cname,cold,it", SINK_CASES)
def test_sinkpit(golden, S, cname, cold, it):
    g = golden("criterion")
    e = t(g["sink{}_est".format(S)]).clone().requires_grad_(True)
    tgt = t(g["sink{}_tgt".format(S)])
    key = "sink{}_{}_{}_{}".format(S, cname, cold, it)
    loss_b, P = CO.sinkpit(make(cname), e, tgt, coldness=cold, iteration=it, batch_mean=False)
    loss_b.mean().backward()
    assert torch.equal(P.argmax(dim=2), t(g[key + "_pattern"]))
    assert_close(loss_b, g[key + "_loss_b"], 1e-5, key)
    assert_close(loss_b.mean(), g[key + "_loss"], 1e-5, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")
