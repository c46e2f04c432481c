#!/usr/bin/env python3
"""Golden vectors for the two data formats either side of the hot path, produced by the
REFERENCE's own code in the build container (TEST INFRASTRUCTURE; the reference never travels):

  G9  tests/golden/g9_save_wav.npz        utils/audio.py:save_wav (float waveform -> int16 .wav), norm False / True, read back
                                          from the files the reference wrote
  G10 tests/golden/g10_dict_embed.{idx,data} + g10_pinyin_encoder.pkl
                                          a four-item ``dict_embed`` dataset written by the reference's own
                                          utils/indexed_datasets.py:IndexedDatasetBuilder with items in the binarizer's
                                          layout (torch tensors for key / value, the SAME tensor object for both, as
                                          data_gen/tts/binarizer_zh.py:232-234 does)
Inputs come from tests/golden_cases.py (seeded)."""
import os
import pickle
import sys
import tempfile

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (stub finder for the third-party imports that are absent offline)


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    import torch
    from scipy.io import wavfile
    import golden_cases as gc
    from utils.audio import save_wav
    from utils.indexed_datasets import IndexedDatasetBuilder, IndexedDataset

    # ---- G9
    lens, wavs = gc.g9_wavs()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for i, w in enumerate(wavs):
            for norm in (False, True):
                p = os.path.join(d, f"u{i}_{int(norm)}.wav")
                save_wav(w.copy(), p, 22050, norm=norm)          # save_wav scales its argument in place
                sr, pcm = wavfile.read(p)
                assert sr == 22050 and pcm.dtype == np.int16
                out[f"u{i}.norm{int(norm)}"] = pcm
    np.savez_compressed(os.path.join(mg.OUT, "g9_save_wav.npz"), **out)
    print("G9", {k: (v.shape, int(v.min()), int(v.max())) for k, v in out.items()})

    # ---- G10
    words, items, pinyin_encoder = gc.g10_entries()
    base = os.path.join(mg.OUT, "g10_dict_embed")
    b = IndexedDatasetBuilder(base)
    for it in items:
        t = torch.from_numpy(it["key"])
        b.add_item({"tokens_gloss": it["tokens_gloss"], "key": t, "key_map": it["key_map"], "value": t,
                    "pinyin": it["pinyin"], "pinyin_map": it["pinyin_map"]})
    b.finalize()
    with open(os.path.join(mg.OUT, "g10_pinyin_encoder.pkl"), "wb") as f:
        pickle.dump(pinyin_encoder, f)
    ds = IndexedDataset(base)
    assert len(ds) == len(items) and torch.equal(ds[2]["key"], torch.from_numpy(items[2]["key"]))
    print("G10 words", words, "L", [it["key"].shape[0] for it in items], "bytes", os.path.getsize(base + ".data"))


if __name__ == "__main__":
    main()
