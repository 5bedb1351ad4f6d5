"""speech_amd/dataset.py against what the reference's datasets returned for the same synthetic corpus
(tests/golden/dataset.npz from tools/gen_golden_dataset.py; the reference read the files through scipy): windowing,
segment IDs, shapes and sample values, bit-exact.  Plus the RIFF reader's edge cases and the wave writer."""
import os
import struct

import numpy as np
import pytest
import torch
from scipy.io import wavfile

import dataset_cases as cases
from speech_amd import dataset as D


def checksum(t):
    f = t.double().flatten()
    idx = torch.linspace(0, max(f.numel() - 1, 0), 8).long()
    return np.concatenate([[f.numel(), f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    return cases.write_corpus(str(tmp_path_factory.mktemp("corpus")))


@pytest.mark.parametrize("name", list(cases.TRAIN_CASES))
def test_train_windows_match_reference(golden, corpus, name):
    g = golden("dataset")
    wav_root, list_path = corpus
    ds = D.WaveTrainDataset(wav_root, list_path, **cases.TRAIN_CASES[name])
    full = D.WaveDataset(wav_root, list_path, **cases.TRAIN_CASES[name])
    assert len(ds) == int(g[name + "_len"])
    assert [full[i][2] for i in range(len(full))] == g[name + "_ids"].tolist()
    for i in range(len(ds)):
        mix, src = ds[i]
        assert list(mix.shape) + list(src.shape) == g[name + "_shapes"][i].tolist()
        assert np.array_equal(checksum(mix), g[name + "_mix"][i]), (name, i)
        assert np.array_equal(checksum(src), g[name + "_src"][i]), (name, i)
        assert mix.dtype == torch.float32 and src.dtype == torch.float32


@pytest.mark.parametrize("name", list(cases.EVAL_CASES))
def test_eval_items_match_reference(golden, corpus, name):
    g = golden("dataset")
    wav_root, list_path = corpus
    ds = D.WaveEvalDataset(wav_root, list_path, **cases.EVAL_CASES[name])
    assert len(ds) == int(g[name + "_len"])
    for i in range(len(ds)):
        mix, src, ID = ds[i]
        assert ID == g[name + "_ids"][i]
        assert list(mix.shape) + list(src.shape) == g[name + "_shapes"][i].tolist()
        assert np.array_equal(checksum(mix), g[name + "_mix"][i])
        assert np.array_equal(checksum(src), g[name + "_src"][i])


def test_loaders_match_reference(golden, corpus):
    g = golden("dataset")
    wav_root, list_path = corpus
    ds = D.WaveTrainDataset(wav_root, list_path, **cases.TRAIN_CASES["train_default"])
    mix, src = next(iter(D.TrainDataLoader(ds, batch_size=3, shuffle=False)))
    assert list(mix.shape) + list(src.shape) == g["loader_train_shapes"].tolist()
    assert np.array_equal(checksum(mix), g["loader_train_mix"])
    ev = D.WaveTestDataset(wav_root, list_path, max_samples=40000)
    mix, src, ids = next(iter(D.TestDataLoader(ev, batch_size=1, shuffle=False)))
    assert list(mix.shape) + list(src.shape) == g["loader_test_shapes"].tolist()
    assert ids == g["loader_test_ids"].tolist()
    with pytest.raises(AssertionError):
        D.EvalDataLoader(ev, batch_size=2)


def test_wav_reader_edges(tmp_path):
    p = os.path.join(str(tmp_path), "x.wav")
    x = (np.arange(-50, 50, dtype=np.int16) * 300).astype(np.int16)
    stereo = np.stack([x, -x], axis=1)
    wavfile.write(p, 8000, stereo)
    w, rate = D.load_wav(p)
    assert rate == 8000 and tuple(w.shape) == (2, 100) and D.wav_frames(p) == (100, 8000)
    assert torch.equal(w, torch.from_numpy(stereo.T.astype(np.float32) / 32768.0))
    w2, _ = D.load_wav(p, frame_offset=90, num_frames=50)          # runs past the end: shorter, not padded
    assert tuple(w2.shape) == (2, 10) and torch.equal(w2, w[:, 90:])
    assert tuple(D.load_wav(p, frame_offset=100)[0].shape) == (2, 0)
    # an extra chunk before 'data' (odd-sized, padded) and 32-bit float samples
    q = os.path.join(str(tmp_path), "f.wav")
    f32 = np.linspace(-1, 1, 7, dtype="<f4")
    with open(q, "wb") as f:
        body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32) + \
               b"LIST" + struct.pack("<I", 3) + b"abc\0" + b"data" + struct.pack("<I", 28) + f32.tobytes()
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    w3, rate3 = D.load_wav(q)
    assert rate3 == 16000 and torch.equal(w3[0], torch.from_numpy(f32.astype(np.float32)))
    with open(q, "r+b") as f:                                       # 8-bit PCM is refused, loudly
        f.seek(20); f.write(struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8))
    with pytest.raises(NotImplementedError):
        D.load_wav(q)
    with open(q, "wb") as f:
        f.write(b"not a wave file at all")
    with pytest.raises(ValueError):
        D.load_wav(q)


def test_wave_writer_matches_scipy(tmp_path):
    x = torch.from_numpy(cases.pcm(1234, 77).astype(np.float32) / 32768.0)
    a, b = os.path.join(str(tmp_path), "a.wav"), os.path.join(str(tmp_path), "b.wav")
    D.save_wav(a, x.unsqueeze(0), 8000)
    wavfile.write(b, 8000, cases.pcm(1234, 77))
    assert open(a, "rb").read() == open(b, "rb").read()
    assert torch.equal(D.load_wav(a)[0][0], x)
