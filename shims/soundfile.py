"""Minimal `soundfile` stand-in: write() through scipy (base_inferencer.py:160)."""
import numpy as np


def write(path, data, samplerate, **_kw):
    from scipy.io import wavfile
    wavfile.write(path, int(samplerate), np.asarray(data, dtype=np.float32))
