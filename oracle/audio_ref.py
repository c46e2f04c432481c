"""ORACLE (test infrastructure, never shipped): CPU restatement of the sample conversion of ``save_wav``
(utils/audio.py:11-16): optional peak normalisation, ``wav * 32767`` in float32, truncating cast to int16
(numpy ``astype``).  Pinned by tests/golden/g9_save_wav.npz, which holds what the reference's own save_wav wrote
(oracle/make_golden_io.py).  Only tests/ may import this module."""
import numpy as np


def save_wav_pcm(wav, norm=False):
    wav = np.array(wav, dtype=np.float32, copy=True)        # utils/audio.py:11 receives the float32 vocoder output
    if norm:
        wav = wav / np.abs(wav).max()                       # :12-13
    wav *= 32767                                            # :14 (in place, float32)
    return wav.astype(np.int16)                             # :16
