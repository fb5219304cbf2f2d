"""Minimal `librosa` stand-in for the reference CLI: util.find_files, load (wav via scipy), stft/istft names."""
import os

import numpy as np

from . import util  # noqa: F401


def load(path, sr=None, mono=True, **_kw):
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max)
    data = data.astype(np.float32)
    if data.ndim > 1 and mono:
        data = data.mean(axis=1)
    if sr is not None and sr != rate:
        raise NotImplementedError("shim librosa.load does not resample")
    return data, rate


def stft(*a, **k):
    raise NotImplementedError("shim: the FullSubNet+ inferencer uses torch.stft")


def istft(*a, **k):
    raise NotImplementedError("shim: the FullSubNet+ inferencer uses torch.istft")
