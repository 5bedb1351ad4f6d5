"""Pins the CPU oracle to outputs of the reference itself (tests/golden, made by tools/gen_golden.py)."""
import numpy as np
import pytest
import torch

import cases
import convtasnet_oracle as O
from conftest import assert_close

TOL = 1e-5   # CPU restatement vs reference, both torch fp32


def test_norm_golden(golden):
    g = golden("norm")
    inp = cases.norm_inputs()
    gy = inp["gy"]
    for kind in ("gln", "cln"):
        sd = {k: v.clone().requires_grad_(True) for k, v in cases.norm_params(kind).items()}
        x = inp[kind].clone().requires_grad_(True)
        if kind == "gln":
            y = O.gln(x, sd["norm.weight"], sd["norm.bias"])
            y2 = O.gln_manual(x, sd["norm.weight"], sd["norm.bias"])
            assert_close(y2, g["gln_y"], TOL, "gln_manual")
        else:
            y = O.cln(x, sd["gamma"], sd["beta"])
        y.backward(gy)
        assert_close(y, g[kind + "_y"], TOL, kind + " y")
        assert_close(x.grad, g[kind + "_dx"], 5e-5, kind + " dx")
        for k, p in sd.items():
            assert_close(p.grad, g["{}_d_{}".format(kind, k)], 5e-5, kind + " d" + k)
    sd = cases.norm_params("gln")
    assert_close(O.gln(torch.zeros(1, 8, 50), sd["norm.weight"], sd["norm.bias"]), g["gln_zero_y"], TOL, "gln zero")
    sd = cases.norm_params("cln")
    assert_close(O.cln(inp["x4"], sd["gamma"], sd["beta"]), g["cln4_y"], TOL, "cln 4-D")


@pytest.mark.parametrize("name,dil,causal,dual", cases.RESBLOCK_CASES)
def test_resblock_golden(golden, name, dil, causal, dual):
    g = golden("resblock")
    sd = {k: v.clone().requires_grad_(True) for k, v in cases.resblock_state(causal, dual).items()}
    x, g_skip, g_out = cases.resblock_inputs(dual)
    x = x.clone().requires_grad_(True)
    out, skip = O.residual_block(x, sd, "", dil, causal, dual)
    loss = (skip * g_skip).sum()
    if dual:
        loss = loss + (out * g_out).sum()
        assert_close(out, g[name + "_out"], TOL, "out")
    else:
        assert out is None
    loss.backward()
    assert_close(skip, g[name + "_skip"], TOL, "skip")
    assert_close(x.grad, g[name + "_dx"], 5e-5, "dx")
    for k, p in sd.items():
        assert_close(p.grad, g["{}_d_{}".format(name, k)], 5e-5, "d" + k)


@pytest.mark.parametrize("name", ["rand2", "rand3", "tie", "swap", "silent"])
def test_pit_golden(golden, name):
    g = golden("pit")
    est = torch.from_numpy(g[name + "_est"]).clone().requires_grad_(True)
    tgt = torch.from_numpy(g[name + "_tgt"])
    loss_b, perm_b = O.pit_neg_sisdr(est, tgt, batch_mean=False)
    loss, perm = O.pit_neg_sisdr(est, tgt)
    loss.backward()
    assert np.array_equal(perm.numpy(), g[name + "_perm"])
    assert np.array_equal(perm_b.numpy(), g[name + "_perm"])
    assert_close(loss_b, g[name + "_loss_b"], TOL, "loss_b")
    assert_close(loss, g[name + "_loss"], TOL, "loss")
    assert_close(est.grad, g[name + "_dest"], 5e-5, "dest")
    if name == "tie":
        assert perm.tolist() == [[0, 1], [0, 1]]
    if name == "swap":
        assert perm.tolist() == [[1, 0], [1, 0]]


def _e2e(golden, name, cfg, full):
    g = golden(name)
    batch, T = int(g["batch"]), int(g["T"])
    sd = cases.e2e_state(cfg)
    mixture, sources = cases.e2e_batch(cfg, batch, T, 0)
    loss, perm, out, grads = O.loss_and_grads(mixture, sources, sd, cfg)
    assert np.array_equal(perm.numpy(), g["perm"])
    assert_close(loss, g["loss"], TOL, "loss")
    assert_close(out[..., ::8], g["out"], 2e-5, "out")
    for k, v in grads.items():
        if full:
            assert_close(v, g["grad." + k], 1e-4, "grad " + k)
        else:
            assert_close(cases.checksum(v), g["gradsum." + k], 1e-4, "gradsum " + k)
    return g, sd


def test_e2e_small_golden_and_two_steps(golden):
    cfg = O.CONFIG_SMALL
    g, sd = _e2e(golden, "e2e_small", cfg, True)
    opt = O.AdamState(sd)
    for step in range(2):
        mixture, sources = cases.e2e_batch(cfg, int(g["batch"]), int(g["T"]), step)
        loss, _ = O.train_step(mixture, sources, sd, cfg, opt)
        if step == 1:
            assert_close(loss, g["loss_step1"], 2e-5, "loss step 1")
        for k, v in sd.items():
            assert_close(v, g["param_step{}.{}".format(step + 1, k)], 2e-5, "param {} step {}".format(k, step + 1))


def test_e2e_small_causal3_golden(golden):
    cfg = dict(O.CONFIG_SMALL, causal=True, n_sources=3, enc_nonlinear="relu")
    _e2e(golden, "e2e_small_causal3", cfg, True)


def test_e2e_paper_golden(golden):
    _e2e(golden, "e2e_paper", O.CONFIG_PAPER, False)


def test_e2e_paper_causal3_golden(golden):
    _e2e(golden, "e2e_paper_causal3", O.CONFIG_CAUSAL3, False)


def test_e2e_paper_causal3_full_length_golden(golden):
    _e2e(golden, "e2e_paper_causal3_full", O.CONFIG_CAUSAL3, False)


def test_parameter_shapes_paper():
    shapes = O.parameter_shapes(O.CONFIG_PAPER)
    assert len(shapes) == 343
    assert sum(int(np.prod(s)) for s in shapes.values()) == 4984881
    assert sum(int(np.prod(s)) for s in O.parameter_shapes(O.CONFIG_CAUSAL3).values()) == 5050929
