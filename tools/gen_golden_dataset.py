"""Generate tests/golden/dataset.npz by running the REFERENCE's wsj0-mix datasets (egs/wsj0-mix/common/src/dataset.py)
on synthetic 16-bit wave files.  Runs only in the build container (needs /root/reference).

torchaudio is not installed: the module registered in its place implements `load` with scipy.io.wavfile (int16 / 32768,
frame_offset / num_frames slicing -- torchaudio's documented behaviour for 16-bit PCM), independent of the reader in
speech_amd/dataset.py that the fixture is there to check.  Wave files are rebuilt by the test from the same seeds
(tests/dataset_cases.py), only what the reference returned is stored.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_dataset.py
"""
import os
import shutil
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/egs/wsj0-mix/common/src")
sys.path.insert(0, "/root/reference/src")
sys.path.insert(2, ROOT)
sys.path.insert(3, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy.io import wavfile  # noqa: E402


def _load(path, frame_offset=0, num_frames=-1):
    rate, data = wavfile.read(path)
    assert data.dtype == np.int16
    data = data.reshape(len(data), -1)
    stop = len(data) if num_frames < 0 else frame_offset + num_frames
    return torch.from_numpy(data[frame_offset:stop].T.astype(np.float32) / 32768.0), rate


_ta = types.ModuleType("torchaudio")
_ta.load = _load
sys.modules.setdefault("torchaudio", _ta)

import dataset as ref  # noqa: E402  (reference)
import dataset_cases as cases  # noqa: E402  (tests/dataset_cases.py)


def checksum(t):
    f = t.double().flatten()
    idx = torch.linspace(0, max(f.numel() - 1, 0), 8).long()
    return np.concatenate([[f.numel(), f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


def main():
    tmp = tempfile.mkdtemp()
    d = {"meta": np.array("reference datasets run on torch {}, wave files read with scipy {}".format(
        torch.__version__, __import__("scipy").__version__))}
    try:
        wav_root, list_path = cases.write_corpus(tmp)
        for name, kw in cases.TRAIN_CASES.items():
            ds = ref.WaveTrainDataset(wav_root, list_path, **kw)
            full = ref.WaveDataset(wav_root, list_path, **kw)
            d[name + "_len"] = np.array(len(ds))
            d[name + "_ids"] = np.array([full[i][2] for i in range(len(full))])
            d[name + "_mix"] = np.stack([checksum(ds[i][0]) for i in range(len(ds))])
            d[name + "_src"] = np.stack([checksum(ds[i][1]) for i in range(len(ds))])
            d[name + "_shapes"] = np.array([list(ds[i][0].shape) + list(ds[i][1].shape) for i in range(len(ds))])
        for name, kw in cases.EVAL_CASES.items():
            ds = ref.WaveEvalDataset(wav_root, list_path, **kw)
            d[name + "_len"] = np.array(len(ds))
            d[name + "_ids"] = np.array([ds[i][2] for i in range(len(ds))])
            d[name + "_mix"] = np.stack([checksum(ds[i][0]) for i in range(len(ds))])
            d[name + "_src"] = np.stack([checksum(ds[i][1]) for i in range(len(ds))])
            d[name + "_shapes"] = np.array([list(ds[i][0].shape) + list(ds[i][1].shape) for i in range(len(ds))])
        # one batch of the reference's loaders
        ds = ref.WaveTrainDataset(wav_root, list_path, **cases.TRAIN_CASES["train_default"])
        mix, src = next(iter(ref.TrainDataLoader(ds, batch_size=3, shuffle=False)))
        d["loader_train_shapes"] = np.array(list(mix.shape) + list(src.shape))
        d["loader_train_mix"] = checksum(mix)
        ev = ref.WaveTestDataset(wav_root, list_path, max_samples=40000)
        mix, src, ids = next(iter(ref.TestDataLoader(ev, batch_size=1, shuffle=False)))
        d["loader_test_shapes"] = np.array(list(mix.shape) + list(src.shape))
        d["loader_test_ids"] = np.array(ids)
    finally:
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dataset.npz"), **d)
    print({k: (v.shape, v if v.size < 8 else "") for k, v in d.items()})


if __name__ == "__main__":
    main()
