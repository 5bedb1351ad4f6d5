"""The criterion variants on the HIP kernels (speech_amd.criterion: SDR / NegSDR, PIT over them and over the 'sum'
reduction, one-and-rest PIT, Sinkhorn PIT) against the reference's own outputs (tests/golden/criterion.npz): losses
within 1e-4, chosen indices bit-exact, gradients within 1e-4; and against the float64 oracle at the bench's signal
length."""
import pytest
import torch
import torch.nn as nn

import criterion_oracle as CO
from conftest import assert_close
from criterion_cases import SDR_CASES, PIT_CASES, ORPIT_CASES, SINK_CASES

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def crit(cname, reduction="mean"):
    from speech_amd.criterion import sdr as C
    return getattr(C, cname)(reduction=reduction)


def t(a):
    return torch.from_numpy(a).to(dev())


@pytest.mark.parametrize("cname,red,bm", SDR_CASES)
def test_sdr_criteria(golden, cname, red, bm):
    g = golden("criterion")
    e = t(g["sdr_est"]).requires_grad_(True)
    loss = crit(cname, red)(e, t(g["sdr_tgt"]), batch_mean=bm)
    loss.sum().backward()
    key = "sdr_{}_{}_{}".format(cname, red, int(bm))
    assert tuple(loss.shape) == g[key + "_loss"].shape
    assert_close(loss, g[key + "_loss"], 1e-4, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")
    assert crit(cname).maximize == (cname == "SDR")


def test_sdr_input_ranks(golden):
    g = golden("criterion")
    e = t(g["sdr_est"])[:, 0].contiguous().requires_grad_(True)
    loss = crit("NegSDR")(e, t(g["sdr_tgt"])[:, 0], batch_mean=False)
    loss.sum().backward()
    assert_close(loss, g["sdr_rank2_loss"], 1e-4, "2-D input")
    assert_close(e.grad, g["sdr_rank2_dest"], 1e-4, "2-D input grad")
    e = t(g["sdr4_est"]).requires_grad_(True)                       # 6 rows: two kernel calls
    loss = crit("NegSDR")(e, t(g["sdr4_tgt"]), batch_mean=False)
    loss.sum().backward()
    assert tuple(loss.shape) == g["sdr4_loss"].shape
    assert_close(loss, g["sdr4_loss"], 1e-4, "4-D input")
    assert_close(e.grad, g["sdr4_dest"], 1e-4, "4-D input grad")
    with pytest.raises(ValueError):
        crit("NegSDR", reduction="median")


@pytest.mark.parametrize("S,cname,red", PIT_CASES)
def test_pit_over_other_criteria(golden, S, cname, red):
    from speech_amd.criterion.pit import PIT1d
    g = golden("criterion")
    e = t(g["pit{}_est".format(S)]).requires_grad_(True)
    tgt = t(g["pit{}_tgt".format(S)])
    key = "pit{}_{}_{}".format(S, cname, red)
    pit = PIT1d(crit(cname, red), S)
    loss_b, perm = pit(e, tgt, batch_mean=False)
    loss, _ = pit(e, tgt)
    loss.backward()
    assert perm.dtype == torch.int64 and torch.equal(perm.cpu(), torch.from_numpy(g[key + "_perm"]))
    assert_close(loss_b, g[key + "_loss_b"], 1e-4, key)
    assert_close(loss, g[key + "_loss"], 1e-4, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")


def test_pit_exact_tie_keeps_first_permutation():
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.utils import synth
    tgt = synth.normal((2, 3, 500), 7, std=0.1)
    tgt[:, 1] = tgt[:, 0]
    tgt[:, 2] = tgt[:, 0]
    est = synth.normal((2, 3, 500), 8, std=0.1)
    est[:, 1] = est[:, 0]
    est[:, 2] = est[:, 0]
    for cname in ("NegSDR", "SDR"):
        _, perm = PIT1d(crit(cname), 3)(est.to(dev()), tgt.to(dev()))
        assert perm.cpu().tolist() == [[0, 1, 2], [0, 1, 2]]


@pytest.mark.parametrize("cname", ORPIT_CASES)
def test_orpit(golden, cname):
    from speech_amd.criterion.pit import ORPIT
    g = golden("criterion")
    e = t(g["orpit_est"]).requires_grad_(True)
    tgt = t(g["orpit_tgt"])
    key = "orpit_" + cname
    orpit = ORPIT(crit(cname))
    loss_b, idx = orpit(e, tgt, batch_mean=False)
    loss, _ = orpit(e, tgt)
    loss.backward()
    assert torch.equal(idx.cpu(), torch.from_numpy(g[key + "_idx"]))
    assert_close(loss_b, g[key + "_loss_b"], 1e-4, key)
    assert_close(loss, g[key + "_loss"], 1e-4, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")
    with pytest.raises(AssertionError):
        orpit(tgt, tgt)                                              # input must hold exactly 'one' and 'rest'


def test_orpit_packed_targets(golden):
    from speech_amd.criterion.pit import ORPIT
    g = golden("criterion")
    tgt = t(g["orpit_tgt"])
    lens = [int(n) for n in g["orpit_packed_lens"]]
    packed = nn.utils.rnn.pack_sequence([tgt[b, :n] for b, n in enumerate(lens)], enforce_sorted=True)
    e = t(g["orpit_packed_est"]).requires_grad_(True)
    orpit = ORPIT(crit("NegSISDR"))
    loss_b, idx = orpit(e, packed, batch_mean=False)
    loss, _ = orpit(e, packed)
    loss.backward()
    assert torch.equal(idx.cpu(), torch.from_numpy(g["orpit_packed_idx"]))
    assert_close(loss_b, g["orpit_packed_loss_b"], 1e-4, "packed ORPIT")
    assert_close(loss, g["orpit_packed_loss"], 1e-4, "packed ORPIT")
    assert_close(e.grad, g["orpit_packed_dest"], 1e-4, "packed ORPIT grad")


@pytest.mark.parametrize("S,cname,cold,it", SINK_CASES)
def test_sinkpit(golden, S, cname, cold, it):
    from speech_amd.criterion.pit import SinkPIT
    g = golden("criterion")
    e = t(g["sink{}_est".format(S)]).requires_grad_(True)
    tgt = t(g["sink{}_tgt".format(S)])
    key = "sink{}_{}_{}_{}".format(S, cname, cold, it)
    sink = SinkPIT(crit(cname), n_sources=S, coldness=cold, iteration=it)
    loss_b, pattern = sink(e, tgt, batch_mean=False)
    loss, _ = sink(e, tgt)
    loss.backward()
    assert torch.equal(pattern.cpu(), torch.from_numpy(g[key + "_pattern"]))
    assert_close(loss_b, g[key + "_loss_b"], 1e-4, key)
    assert_close(loss, g[key + "_loss"], 1e-4, key)
    assert_close(e.grad, g[key + "_dest"], 1e-4, key + " grad")


def test_sinkpit_generic_criterion_path_matches_pairwise_path():
    """A criterion the pairwise kernels do not know (a subclass) takes the reference's expand-and-call route
    (pit.py:169-173); both routes agree."""
    from speech_amd.criterion.pit import SinkPIT
    from speech_amd.criterion.sdr import NegSISDR
    from speech_amd.utils import synth

    class Other(NegSISDR):
        pass

    tgt = synth.normal((2, 3, 4000), 17, std=0.1).to(dev())
    est0 = (0.7 * tgt[:, [2, 0, 1]] + synth.normal((2, 3, 4000), 18, std=0.05).to(dev()))
    res = []
    for c in (NegSISDR(), Other()):
        e = est0.clone().requires_grad_(True)
        loss, pattern = SinkPIT(c, n_sources=3)(e, tgt)
        loss.backward()
        res.append((loss.detach(), pattern, e.grad))
    assert torch.equal(res[0][1], res[1][1])
    assert_close(res[0][0], res[1][0], 1e-5, "loss")
    assert_close(res[0][2], res[1][2], 1e-4, "grad")


@pytest.mark.parametrize("S", [2, 3])
def test_variants_at_bench_length_vs_float64(S):
    """32000-sample signals (the bench's utterances): SDR values, Sinkhorn PIT loss and gradients against the float64
    oracle."""
    from speech_amd.criterion.pit import SinkPIT, PIT1d
    from speech_amd.utils import synth
    B, T = 3, 32000
    tgt = synth.normal((B, S, T), 61 + S, std=0.1)
    est = 0.6 * tgt.flip(1) + synth.normal((B, S, T), 71 + S, std=0.04)
    e64 = est.double().requires_grad_(True)
    ref_c = CO.Criterion(CO.sdr, negate=True)
    ref_loss, ref_P = CO.sinkpit(ref_c, e64, tgt.double(), coldness=0.5, iteration=10)
    ref_loss.backward()
    e = est.to(dev()).requires_grad_(True)
    loss, pattern = SinkPIT(crit("NegSDR"), n_sources=S, coldness=0.5)(e, tgt.to(dev()))
    loss.backward()
    assert torch.equal(pattern.cpu(), ref_P.argmax(dim=2))
    assert_close(loss, ref_loss, 1e-5, "Sinkhorn loss")
    assert_close(e.grad, e64.grad, 1e-4, "Sinkhorn grad")
    import convtasnet_oracle as O
    ref_lb, ref_perm = O.pit(ref_c, est.double(), tgt.double(), batch_mean=False)
    lb, perm = PIT1d(crit("NegSDR"), S)(est.to(dev()), tgt.to(dev()), batch_mean=False)
    assert torch.equal(perm.cpu(), ref_perm)
    assert_close(lb, ref_lb, 1e-5, "PIT NegSDR")


def test_orpit_recovers_the_one_from_unsorted_packed_targets():
    """The scenario of the reference's own self-test (src/criterion/pit.py:268-295): integer-valued sources, the input
    holds source 0 and the sum of the others, the target lists the sources in a scrambled order and the batch is packed
    unsorted; ORPIT must name the position of source 0 in every target, with a perfect score."""
    from speech_amd.criterion.pit import ORPIT, ProbPIT
    from speech_amd.utils import synth
    T = 256
    orders = [[2, 1, 0], [0, 1], [0, 2, 1], [1, 0]]
    inputs, targets = [], []
    for b, order in enumerate(orders):
        src = torch.floor(5 * synth.uniform((len(order), T), 90 + b, lo=0.0, hi=1.0))
        inputs.append(torch.stack([src[0], src[1:].sum(dim=0)]))
        targets.append(src[order])
    est = torch.stack(inputs).to(dev())
    packed = nn.utils.rnn.pack_sequence([t_.to(dev()) for t_ in targets], enforce_sorted=False)
    loss_b, idx = ORPIT(crit("SISDR"))(est, packed, batch_mean=False)
    assert idx.cpu().tolist() == [order.index(0) for order in orders]
    assert (loss_b > 100).all()                   # exact reconstruction: SI-SDR limited only by eps
    loss_b, idx = ORPIT(crit("NegSDR"))(est, packed, batch_mean=False)
    assert idx.cpu().tolist() == [order.index(0) for order in orders]
    assert ProbPIT()(est, est) is None            # empty in the reference as well


def test_pair_matrix_shape_sweep_vs_float64():
    """The pairwise SI-SDR / SDR matrix and its full derivative over 1-4 sources and signal lengths from one sample to
    beyond a block's chunk (two samples at least: one sample is always a perfect scaled match), against float64 on the CPU (values 1e-5, gradients 1e-4)."""
    from speech_amd import ops
    from speech_amd.criterion.sdr import PairFn
    from speech_amd.utils import synth
    for n, (B, S, T) in enumerate([(1, 1, 2), (2, 2, 3), (1, 3, 255), (3, 4, 4097), (2, 2, 9000), (5, 1, 513)]):
        tgt = synth.normal((B, S, T), 800 + n, std=0.1)
        est = 0.5 * tgt.flip(1) + synth.normal((B, S, T), 820 + n, std=0.05)
        G = synth.normal((B, S, S), 840 + n)
        for kind, fn in ((ops.PAIR_SISDR, CO.sisdr), (ops.PAIR_SDR, CO.sdr)):
            e64 = est.double().requires_grad_(True)
            ref = torch.stack([torch.stack([fn(e64[:, i], tgt[:, j].double()) for j in range(S)], dim=1)
                               for i in range(S)], dim=1)
            (ref * G.double()).sum().backward()
            e = est.to(dev()).requires_grad_(True)
            pair = PairFn.apply(e, tgt.to(dev()), 1e-12, kind)
            (pair * G.to(dev())).sum().backward()
            what = "kind {} shape {}".format(kind, (B, S, T))
            assert_close(pair, ref, 1e-5, what)
            assert_close(e.grad, e64.grad, 1e-4, what + " grad")
