"""CPU-side checks of the product package: module surface and state_dict keys, error behaviour, the C ABI
(every symbol of include/ctn.h exported, struct mirrors the right size) and a dry run of every schedule with the
real library validating all arguments (launches fail with 'no device' here, which the dry run tolerates)."""
import os
import re
import sys

import pytest
import torch

import cases
import convtasnet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(cfg):
    from speech_amd.models.conv_tasnet import ConvTasNet
    return ConvTasNet(cfg["n_basis"], cfg["kernel_size"], stride=cfg["stride"], enc_basis="trainable",
                      dec_basis="trainable", enc_nonlinear=cfg["enc_nonlinear"],
                      sep_hidden_channels=cfg["sep_hidden_channels"],
                      sep_bottleneck_channels=cfg["sep_bottleneck_channels"],
                      sep_skip_channels=cfg["sep_skip_channels"], sep_kernel_size=cfg["sep_kernel_size"],
                      sep_num_blocks=cfg["sep_num_blocks"], sep_num_layers=cfg["sep_num_layers"], dilated=True,
                      separable=True, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear=cfg["mask_nonlinear"],
                      causal=cfg["causal"], n_sources=cfg["n_sources"])


@pytest.mark.parametrize("cfg,n_params", [(O.CONFIG_SMALL, 60689), (O.CONFIG_PAPER, 4984881),
                                          (O.CONFIG_CAUSAL3, 5050929)])
def test_state_dict_keys_match_reference(cfg, n_params):
    m = build(cfg)
    shapes = O.parameter_shapes(cfg)          # pinned to the reference by test_oracle_golden
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, s in shapes.items():
        assert tuple(sd[k].shape) == tuple(s), k
    assert m.num_parameters == n_params
    assert len(m.fused_parameters()) == len(sd)
    cfgd = m.get_config()
    for key in ("n_basis", "kernel_size", "stride", "enc_basis", "dec_basis", "enc_nonlinear", "window_fn",
                "sep_hidden_channels", "sep_bottleneck_channels", "sep_skip_channels", "sep_kernel_size",
                "sep_num_blocks", "sep_num_layers", "dilated", "separable", "causal", "sep_nonlinear", "sep_norm",
                "mask_nonlinear", "n_sources", "eps", "in_channels", "enc_onesided", "enc_return_complex"):
        assert key in cfgd
    assert m.get_package() == cfgd


def test_build_model_round_trip(tmp_path):
    from speech_amd.models.conv_tasnet import ConvTasNet
    m = build(O.CONFIG_SMALL)
    pkg = m.get_config()
    pkg["state_dict"] = m.state_dict()
    path = os.path.join(str(tmp_path), "last.pth")
    torch.save(pkg, path)
    m2 = ConvTasNet.build_model(path, load_state_dict=True)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_constructor_errors():
    from speech_amd.models.conv_tasnet import ConvTasNet
    from speech_amd.utils.tasnet import choose_layer_norm
    from speech_amd.utils.model import choose_nonlinear
    with pytest.raises(AssertionError):
        ConvTasNet(64, 16, stride=5, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None)
    with pytest.raises(ValueError):
        choose_layer_norm("gLN", 8, causal=True)
    with pytest.raises(NotImplementedError):
        choose_layer_norm("xLN", 8)
    with pytest.raises(ValueError):
        ConvTasNet(64, 16, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None, mask_nonlinear="relu")
    with pytest.raises(AssertionError):
        choose_nonlinear("softmax")
    m = ConvTasNet(64, 16, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None, causal=False)
    assert m.stride == 8
    with pytest.raises(ValueError):
        m.extract_latent(torch.zeros(2, 100))


def test_no_cpu_fallback():
    from speech_amd._lib import CtnError
    from speech_amd.modules.norm import GlobalLayerNorm
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    with pytest.raises(CtnError):
        GlobalLayerNorm(4)(torch.randn(1, 4, 16))
    with pytest.raises(CtnError):
        NegSISDR()(torch.randn(2, 2, 100), torch.randn(2, 2, 100))
    with pytest.raises(CtnError):
        build(O.CONFIG_SMALL)(torch.randn(1, 1, 4000))
    # a criterion without a fused kernel runs the reference's generic permutation loop on any device
    mse = lambda a, b, batch_mean=True: ((a - b) ** 2).mean(dim=(1, 2))
    est = torch.randn(3, 2, 50)
    loss, perm = PIT1d(mse, 2)(est[:, [1, 0]], est, batch_mean=False)
    assert perm.tolist() == [[1, 0]] * 3 and torch.allclose(loss, torch.zeros(3))


def test_abi_exports_every_declared_symbol():
    from speech_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ctn.h")).read()
    declared = set(re.findall(r"\b(ctn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ctn_version().startswith(b"ctn-hip")
    src_newer = [f for f in os.listdir(os.path.join(ROOT, "speech_amd", "csrc")) if f.endswith((".hip", ".h"))
                 and os.path.getmtime(os.path.join(ROOT, "speech_amd", "csrc", f)) > os.path.getmtime(_lib.LIB_PATH)]
    assert not src_newer, "libctn_hip.so is older than {}: run python -m speech_amd.build".format(src_newer)


def test_abi_rejects_bad_arguments():
    from speech_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 4, 8)
    assert lib.ctn_norm_stats(x.data_ptr() + 4, 1, 4, 8, 8, None, x.data_ptr(), None, 0, None) == -1   # misaligned
    assert b"aligned" in lib.ctn_last_error()
    assert lib.ctn_norm_stats(x.data_ptr(), 1, 4, 8, 6, None, x.data_ptr(), None, 0, None) == -1       # ld % 4
    assert lib.ctn_gemm_nn(None, None) == -1
    a = _lib.GemmNTArgs()
    a.M = a.N = a.T = a.batch = 4
    a.kchunk = 33
    assert lib.ctn_gemm_nt(a, None) == -1 and b"kchunk" in lib.ctn_last_error()


@pytest.fixture
def dry_run(monkeypatch):
    """Route the schedules to CPU tensors: the real library still validates every argument, only the kernel
    launch itself fails (no device), which is accepted here."""
    from speech_amd import _lib, ops
    real = _lib.check

    def check(rc, what):
        if rc == -2:
            return
        real(rc, what)
    monkeypatch.setattr(_lib, "check", check)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "_require_cuda", lambda *a: None)
    monkeypatch.setattr(ops, "is_act", lambda x: (x.dim() == 3 and x.dtype == torch.float32 and x.stride(2) == 1
                                                  and x.stride(1) % 4 == 0 and x.stride(0) % 4 == 0
                                                  and x.stride(1) >= x.shape[2] and x.data_ptr() % 16 == 0))


@pytest.mark.parametrize("cfg", [O.CONFIG_SMALL, dict(O.CONFIG_SMALL, causal=True, n_sources=3, enc_nonlinear="relu"),
                                 dict(O.CONFIG_PAPER, sep_num_layers=2)])
def test_schedules_dry_run(dry_run, cfg):
    from speech_amd.criterion.pit import PIT1d
    from speech_amd.criterion.sdr import NegSISDR
    if torch.cuda.is_available():
        pytest.skip("dry run is for GPU-less hosts")
    m = build(cfg)
    mixture, sources = cases.e2e_batch(cfg, 2, 4003, 0)
    out = m(mixture)
    assert out.shape == (2, cfg["n_sources"], 4003)
    loss, perm = PIT1d(NegSISDR(), cfg["n_sources"])(out, sources)
    loss.backward()
    assert perm.shape == (2, cfg["n_sources"]) and perm.dtype == torch.int64
    assert all(p.grad is not None and p.grad.shape == p.shape for p in m.parameters())
    m.zero_grad()
    o, latent = m.extract_latent(mixture)
    assert latent.shape == (2, cfg["n_sources"], cfg["n_basis"], (4003 + 5 - 16) // 8 + 1)
    (o.sum() + latent.sum()).backward()
    assert all(p.grad is not None for p in m.parameters())


def test_encoder_geometry():
    from speech_amd.functional import _enc_geometry
    assert _enc_geometry(32000, 16, 8) == (0, 0, 3999)
    assert _enc_geometry(8003, 16, 8) == (5, 2, 1000)
    assert _enc_geometry(64000, 2, 1) == (0, 0, 63999)


def test_flat_layout_is_aligned_and_disjoint():
    """Every tensor of the flat parameter / gradient buffers starts on a 256-byte boundary (scalars must not
    push the weight matrices off the GEMMs' 16-byte paths) and no two tensors overlap."""
    import torch
    from speech_amd.functional import flat_layout, FLAT_ALIGN
    params = [torch.zeros(s) for s in [(512, 128, 1), (1,), (512,), (1,), (3, 7), (128, 512, 1)]]
    offsets, total = flat_layout(params)
    end = 0
    for p, off in zip(params, offsets):
        assert off % FLAT_ALIGN == 0 and off >= end
        end = off + p.numel()
    assert total >= end and total % FLAT_ALIGN == 0


def test_dprnn_tasnet_surface_matches_reference(golden):
    """state_dict keys, parameter count and config of DPRNN-TasNet against what the reference's model reported."""
    import dprnn_cases
    from speech_amd.models.dprnn_tasnet import DPRNNTasNet
    g = golden("dprnn")
    for name, (cfg, _, _) in dprnn_cases.E2E.items():
        m = DPRNNTasNet(**cfg)
        assert list(m.state_dict().keys()) == g[name + "_keys"].tolist()
        assert m.num_parameters == int(g[name + "_nparams"])
        c = m.get_config()
        for key in ("sep_chunk_size", "sep_hop_size", "sep_num_blocks", "rnn_type", "causal", "n_sources"):
            assert key in c
    with pytest.raises(ValueError):
        DPRNNTasNet(**dict(dprnn_cases.SMALL, mask_nonlinear="tanh"))
    with pytest.raises(ValueError):
        DPRNNTasNet(**dict(dprnn_cases.SMALL, causal=False))(torch.zeros(3, 10))


def test_criterion_variants_surface_and_no_cpu_fallback():
    """SDR / NegSDR (src/criterion/sdr.py:24-114), ORPIT and SinkPIT (src/criterion/pit.py:87-213): constructor
    arguments, the maximize flags, and the rule that the HIP criteria refuse CPU tensors instead of falling back."""
    import inspect
    from speech_amd._lib import CtnError
    from speech_amd.criterion import pit as P, sdr as S
    assert list(inspect.signature(S.NegSDR.__init__).parameters) == ["self", "reduction", "eps"]
    assert list(inspect.signature(P.ORPIT.__init__).parameters) == ["self", "criterion"]
    assert list(inspect.signature(P.SinkPIT.__init__).parameters) == ["self", "criterion", "n_sources", "coldness", "iteration"]
    assert list(inspect.signature(P.sinkpit).parameters) == ["criterion", "input", "target", "n_sources", "coldness",
                                                             "iteration", "batch_mean"]
    assert S.SDR().maximize and not S.NegSDR().maximize and S.NegSDR().eps == 1e-12
    assert P.ORPIT(S.NegSDR()).patterns.tolist() == [[0, 1], [1, 0]]
    with pytest.raises(ValueError):
        S.SDR(reduction="max")
    e, t = torch.randn(2, 2, 64), torch.randn(2, 2, 64)
    with pytest.raises(CtnError):
        S.NegSDR()(e, t)
    with pytest.raises(CtnError):
        P.ORPIT(S.NegSISDR())(e, t)
    # a criterion of the caller's own runs the generic routes on any device
    mse = lambda a, b, batch_mean=True: ((a - b) ** 2).mean(dim=-1)
    loss, P_ = P.sinkpit(mse, e, e[:, [1, 0]], coldness=50.0)
    assert P_.argmax(dim=2).tolist() == [[1, 0], [1, 0]]
    loss, idx = P.ORPIT(mse)(torch.stack([t[:, 1], t[:, 0]], dim=1), t, batch_mean=False)
    assert idx.tolist() == [1, 1] and torch.allclose(loss, torch.zeros(2))


def test_reference_import_names_resolve_through_compat_path():
    """The recipes import top-level `models`, `criterion`, `utils`, `dataset`, `driver`, `adhoc_driver`, `norm`, `conv`
    (egs/wsj0-mix/conv-tasnet/local/train.py:9-14, path.sh:4); with compat/ on PYTHONPATH in place of the reference's
    src directories those statements must load this package's classes."""
    import subprocess
    code = """
import warnings
from utils.utils import set_seed
from dataset import WaveTrainDataset, WaveEvalDataset, TrainDataLoader, EvalDataLoader
from adhoc_driver import AdhocTrainer
from models.conv_tasnet import ConvTasNet
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d
from driver import TrainerBase, TesterBase
from utils.tasnet import choose_layer_norm
from utils.filterbank import choose_filterbank
from utils.model import choose_nonlinear
with warnings.catch_warnings(record=True) as seen:
    warnings.simplefilter("always")
    from norm import GlobalLayerNorm, CumulativeLayerNorm1d
    from conv import DepthwiseSeparableConv1d
assert sum(issubclass(w.category, FutureWarning) for w in seen) == 2
import speech_amd.models.conv_tasnet, speech_amd.criterion.pit, speech_amd.modules.norm, speech_amd.driver
assert ConvTasNet is speech_amd.models.conv_tasnet.ConvTasNet
assert PIT1d is speech_amd.criterion.pit.PIT1d
assert GlobalLayerNorm is speech_amd.modules.norm.GlobalLayerNorm
assert issubclass(AdhocTrainer, TrainerBase) and AdhocTrainer is speech_amd.driver.AdhocTrainer
set_seed(111)
m = ConvTasNet(16, 4, enc_basis='trainable', dec_basis='trainable', enc_nonlinear=None, sep_hidden_channels=8,
               sep_bottleneck_channels=8, sep_skip_channels=8, sep_num_blocks=1, sep_num_layers=2, causal=False)
assert 'separator.tdcn.net.0.net.0.separable_conv1d.depthwise_conv1d.weight' in m.state_dict()
print('compat ok')
"""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat"))
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "compat ok" in r.stdout, r.stdout + r.stderr


def test_framing_helpers_follow_the_reference_arithmetic():
    """frame_padding / waveform_rows / decode_sources (src/models/conv_tasnet.py:127-159): the shortfall to a whole number
    of hops, its smaller half in front; (B, 1, mics, T) as rows per microphone; the framing zeros cut off after decoding."""
    import torch
    from speech_amd.utils.tasnet import frame_padding, waveform_rows, decode_sources
    for T in (100, 101, 117, 32000, 31999):
        for K, S in ((16, 8), (2, 1), (20, 10), (32, 16), (16, 4)):
            left, right = frame_padding(T, K, S)
            pad = (S - (T - K) % S) % S
            assert (left, right) == (pad // 2, pad - pad // 2)
            assert (T + left + right - K) % S == 0
    rows, mics = waveform_rows(torch.zeros(3, 1, 50))
    assert rows.shape == (3, 1, 50) and mics is None
    rows, mics = waveform_rows(torch.arange(3 * 4 * 50.).view(3, 1, 4, 50))
    assert rows.shape == (3, 4, 50) and mics == 4 and float(rows[2, 3, 49]) == 3 * 4 * 50 - 1
    with pytest.raises(ValueError):
        waveform_rows(torch.zeros(3, 50))
    with pytest.raises(AssertionError):
        waveform_rows(torch.zeros(3, 2, 50))
    latent = torch.arange(2 * 3 * 5 * 7.).view(2, 3, 5, 7)
    dec = lambda w: w.sum(dim=1, keepdim=True).repeat_interleave(2, dim=2)      # (B*S, 1, 14)
    out = decode_sources(dec, latent, None, 2, 1)
    assert out.shape == (2, 3, 11)
    assert torch.equal(out[1, 2], latent[1, 2].sum(0).repeat_interleave(2)[2:13])
