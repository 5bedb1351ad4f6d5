"""Generate tests/golden/trainer.npz and tests/golden/trainer_ref_last.pth by running the REFERENCE's trainer on CPU.

Runs only in the build container (needs /root/reference).  Two things are recorded:

* the per-epoch decisions (learning rate, epochs without improvement, best loss, which epochs wrote best.pth, where
  training stopped) of `TrainerBase.run` and of the Conv-TasNet recipe's `AdhocTrainer.run` for scripted sequences
  of validation losses (run_one_epoch is replaced by the script; everything else is the reference's code);
* a real miniature training run of `TrainerBase` (small configuration, 2 epochs of 3 batches, 2 validation
  utterances, clip 5, Adam 1e-3), the checkpoint it wrote (a plain torch.save dictionary of tensors and numbers)
  and the third epoch the reference computes after resuming from that checkpoint.

torchaudio and mir_eval are not installed; the trainer module imports them at the top for wave output and BSS
evaluation, so empty modules are registered (torchaudio.save does nothing).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_trainer.py
"""
import argparse
import copy
import os
import shutil
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/egs/wsj0-mix/conv-tasnet/src")
sys.path.insert(0, "/root/reference/egs/wsj0-mix/common/src")
sys.path.insert(0, "/root/reference/src")
sys.path.insert(3, ROOT)
_ta = types.ModuleType("torchaudio")
_ta.save = lambda *a, **k: None
sys.modules.setdefault("torchaudio", _ta)
_me, _sep = types.ModuleType("mir_eval"), types.ModuleType("mir_eval.separation")
_sep.bss_eval_sources = None
_me.separation = _sep
sys.modules.setdefault("mir_eval", _me)
sys.modules.setdefault("mir_eval.separation", _sep)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.conv_tasnet import ConvTasNet  # noqa: E402  (reference)
from criterion.sdr import NegSISDR  # noqa: E402  (reference)
from criterion.pit import PIT1d  # noqa: E402  (reference)
from driver import TrainerBase  # noqa: E402  (reference)
from adhoc_driver import AdhocTrainer  # noqa: E402  (reference)

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "speech_amd", "utils", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)

SMALL = dict(n_basis=64, kernel_size=16, stride=8, enc_basis='trainable', dec_basis='trainable', enc_nonlinear=None,
             sep_hidden_channels=64, sep_bottleneck_channels=32, sep_skip_channels=32, sep_kernel_size=3,
             sep_num_blocks=2, sep_num_layers=4, dilated=True, separable=True, causal=False, sep_nonlinear='prelu',
             sep_norm=True, mask_nonlinear='sigmoid', n_sources=2)
T = 8000
SCRIPTS = {
    "plateau": [3.0, 2.0, 2.5, 2.6, 2.7, 2.8, 2.9, 1.0, 1.5, 1.6, 1.7],
    "long_plateau": [1.0] + [2.0 + 0.1 * i for i in range(13)],
    "zigzag": [5.0, 6.0, 5.5, 5.6, 5.4, 5.7, 5.8, 5.9, 4.0, 4.0, 4.0, 4.0],
}


def make_args(tmp, epochs, continue_from=None):
    return argparse.Namespace(sample_rate=8000, n_sources=2, max_norm=5, model_dir=os.path.join(tmp, "model"),
                              loss_dir=os.path.join(tmp, "loss"), sample_dir=os.path.join(tmp, "sample"), epochs=epochs,
                              use_cuda=False, continue_from=continue_from, overwrite=True)


def make_model():
    model = ConvTasNet(**SMALL)
    synth.fill_parameters(model, salt=5)
    return model


class ValidSet:
    """What the reference's evaluation loop needs from its loader: iteration and `.dataset` with a length."""

    def __init__(self, items):
        self.items, self.dataset = items, items

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)


def make_loaders():
    train = []
    for i in range(3):
        src = 0.1 * synth.normal((2, 2, T), 700 + i)
        train.append((src.sum(1, keepdim=True), src))
    valid = []
    for i in range(2):
        src = 0.1 * synth.normal((1, 2, T), 800 + i)
        valid.append((src.sum(1, keepdim=True), src, ["utt{}".format(i)]))
    return {"train": train, "valid": ValidSet(valid)}


def checksum(t):
    f = t.detach().double().flatten()
    idx = torch.linspace(0, f.numel() - 1, 16).long()
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def model_checksums(model):
    return np.stack([checksum(p) for p in model.parameters()])


def scripted(cls, script, tmp):
    model = make_model()
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    trainer = cls(model, {"train": [], "valid": ValidSet([])}, PIT1d(NegSISDR(), n_sources=2), optimizer,
                  make_args(tmp, len(script)))
    log = []
    saved_best = []
    orig_save = trainer.save_model

    def save_model(epoch, model_path='./tmp.pth'):
        if model_path.endswith("best.pth"):
            saved_best.append(epoch)
        orig_save(epoch, model_path)

    def run_one_epoch(epoch):
        if epoch > 0:       # state after the previous epoch's decisions
            log.append((optimizer.param_groups[0]['lr'], trainer.no_improvement, trainer.best_loss))
        return 0.0, script[epoch]

    trainer.save_model = save_model
    trainer.run_one_epoch = run_one_epoch
    trainer.run()
    log.append((optimizer.param_groups[0]['lr'], trainer.no_improvement, trainer.best_loss))
    last = torch.load(os.path.join(tmp, "model", "last.pth"), map_location="cpu", weights_only=True)
    return dict(lr=np.array([e[0] for e in log]), no_improvement=np.array([e[1] for e in log]),
                best=np.array([e[2] for e in log]), saved_best=np.array(saved_best), last_epoch=np.array(last['epoch']))


def main():
    d = {"meta": np.array("reference trainer run on torch {} CPU fp32".format(torch.__version__))}
    for name, script in SCRIPTS.items():
        d["script_" + name] = np.array(script)
        for tag, cls in (("base", TrainerBase), ("adhoc", AdhocTrainer)):
            tmp = tempfile.mkdtemp()
            try:
                for k, v in scripted(cls, script, tmp).items():
                    d["{}_{}_{}".format(tag, name, k)] = v
            finally:
                shutil.rmtree(tmp)

    # ---- a real miniature run: 2 epochs, checkpoint, 1 resumed epoch
    tmp = tempfile.mkdtemp()
    try:
        model = make_model()
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
        trainer = TrainerBase(model, make_loaders(), PIT1d(NegSISDR(), n_sources=2), optimizer, make_args(tmp, 2))
        trainer.run()
        d["mini_train_loss2"] = trainer.train_loss.numpy().copy()
        d["mini_valid_loss2"] = trainer.valid_loss.numpy().copy()
        d["mini_params2"] = model_checksums(model)
        d["mini_exp_avg2"] = np.stack([checksum(optimizer.state[p]['exp_avg']) for p in model.parameters()])
        d["mini_exp_avg_sq2"] = np.stack([checksum(optimizer.state[p]['exp_avg_sq']) for p in model.parameters()])
        last = os.path.join(tmp, "model", "last.pth")
        ckpt = torch.load(last, map_location="cpu", weights_only=True)
        d["mini_ckpt_keys"] = np.array(sorted(ckpt.keys()))
        d["mini_optim_group_keys"] = np.array(sorted(ckpt['optim_dict']['param_groups'][0].keys()))
        d["mini_optim_state_keys"] = np.array(sorted(ckpt['optim_dict']['state'][0].keys()))
        shutil.copy(last, os.path.join(OUT, "trainer_ref_last.pth"))

        model3 = make_model()
        optimizer3 = torch.optim.Adam(model3.parameters(), lr=1e-3)
        trainer3 = TrainerBase(model3, make_loaders(), PIT1d(NegSISDR(), n_sources=2), optimizer3,
                               make_args(tmp, 3, continue_from=last))
        trainer3.run()
        d["mini_train_loss3"] = trainer3.train_loss.numpy().copy()
        d["mini_valid_loss3"] = trainer3.valid_loss.numpy().copy()
        d["mini_params3"] = model_checksums(model3)
    finally:
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "trainer.npz"), **d)
    print("wrote trainer.npz and trainer_ref_last.pth;", {k: v.shape for k, v in d.items() if k.startswith("mini")})
    print("losses", d["mini_train_loss3"], d["mini_valid_loss3"])


if __name__ == "__main__":
    main()
