"""A synthetic wsj0-mix style corpus (16-bit PCM, 8 kHz) shared by tools/gen_golden_dataset.py and the dataset tests:
the same seeds give the same wave files on both sides."""
import os

import numpy as np
from scipy.io import wavfile

from speech_amd.utils import synth

# utterance -> frames: two windows of 32000 at hop 16000 fit in 50000 (one at hop 24000), three in 70001; the last
# utterance is shorter than a window and yields none
UTTERANCES = {"utt_a": 50000, "utt_b": 70001, "utt_c": 31999, "utt_d": 32000}
TRAIN_CASES = {"train_default": dict(samples=32000, overlap=None, n_sources=2),
               "train_hop24000": dict(samples=32000, overlap=8000, n_sources=2),
               "train_3src": dict(samples=16000, overlap=0, n_sources=3)}
EVAL_CASES = {"eval_full": dict(max_samples=None, n_sources=2), "eval_40000": dict(max_samples=40000, n_sources=3)}
RATE = 8000


def pcm(frames, seed):
    x = synth.normal((frames,), seed).numpy() * 0.1
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)


def write_corpus(root):
    """Writes <root>/wav/{mix,s1,s2,s3}/<ID>.wav and <root>/list.txt; returns (wav_root, list_path)."""
    wav_root = os.path.join(root, "wav")
    for sub in ("mix", "s1", "s2", "s3"):
        os.makedirs(os.path.join(wav_root, sub), exist_ok=True)
    for n, (ID, frames) in enumerate(UTTERANCES.items()):
        srcs = [pcm(frames, 9000 + 10 * n + k) for k in range(3)]
        for k, s in enumerate(srcs):
            wavfile.write(os.path.join(wav_root, "s{}".format(k + 1), ID + ".wav"), RATE, s)
        mix = np.clip(srcs[0].astype(np.int32) + srcs[1] + srcs[2], -32768, 32767).astype(np.int16)
        wavfile.write(os.path.join(wav_root, "mix", ID + ".wav"), RATE, mix)
    list_path = os.path.join(root, "list.txt")
    with open(list_path, "w") as f:
        f.write("\n".join(UTTERANCES.keys()) + "\n")
    return wav_root, list_path
