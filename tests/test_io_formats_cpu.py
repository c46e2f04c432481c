"""CPU tests of the two data formats either side of the hot path against fixtures the REFERENCE's
own code wrote (oracle/make_golden_io.py): the ``dict_embed`` indexed dataset -> resident-table conversion, and the
save_wav sample conversion."""
import os
import subprocess
import sys

import numpy as np

import golden_cases as gc
from dict_tts_amd import dict_embed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dict_embed_dataset_to_table(golden_dir):
    """G10: a dataset written by the reference's IndexedDatasetBuilder with binarizer-layout items converts to exactly the
    ragged arrays dtts_dict_table_upload takes"""
    words, items, pinyin_encoder = gc.g10_entries()
    base = os.path.join(golden_dir, "g10_dict_embed")
    raw = dict_embed.read_indexed_dataset(base)
    assert len(raw) == len(items) == 4
    assert set(raw[1]) == {"tokens_gloss", "key", "key_map", "value", "pinyin", "pinyin_map"}   # binarizer_zh.py:301-307
    t = dict_embed.table_from_dict_embed(base, os.path.join(golden_dir, "g10_pinyin_encoder.pkl"))
    assert t["values"] is None                       # the reference stores value == key (binarizer_zh.py:232-233)
    assert t["tok_off"].dtype == np.int32 and t["tok_off"].tolist() == np.cumsum([0] + [it["key"].shape[0] for it in items]).tolist()
    assert t["pin_off"].tolist() == np.cumsum([0] + [len(it["pinyin"]) for it in items]).tolist()
    for i, it in enumerate(items):
        a, b = t["tok_off"][i], t["tok_off"][i + 1]
        assert np.array_equal(t["keys"][a:b], it["key"]) and t["keys"].dtype == np.float32
        assert t["key_map"][a:b].tolist() == it["key_map"]
        p, q = t["pin_off"][i], t["pin_off"][i + 1]
        assert t["pinyin"][p:q].tolist() == [pinyin_encoder.index(s) for s in it["pinyin"]]   # dataset_utils.py:322
        assert t["pinyin_map"][p:q].tolist() == it["pinyin_map"]
    # item 0 is the binarizer's entry for a word outside zh-dict.json (binarizer_zh.py:250-259)
    assert t["L"][0] == 3 and not t["keys"][:3].any() and t["key_map"][:3].tolist() == [0, 1, 0] and t["pinyin"][0] == 0
    assert int(t["pinyin_map"].max()) == 3           # the three-sense heteronym
    # the reference's per-character lookup (dataset_utils.py:313-318): unknown words -> entry 2
    ids = dict_embed.entry_ids_for_words(["b", "zz", "d"], {"a": 0, "b": 1, "c": 2, "d": 3}, 4)
    assert ids.dtype == np.int32 and ids.tolist() == [1, 2, 3]


def test_dict_embed_cli(golden_dir, tmp_path):
    out = tmp_path / "table.npz"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dict_embed_to_table.py"),
                        os.path.join(golden_dir, "g10_dict_embed"), os.path.join(golden_dir, "g10_pinyin_encoder.pkl"), str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    z = np.load(out)
    assert set(z.files) == {"tok_off", "keys", "key_map", "pin_off", "pinyin", "pinyin_map"} and z["keys"].shape == (48, 768)


def test_oracle_save_wav_vs_reference_golden(golden_dir):
    """G9: the oracle's sample conversion equals what the reference's save_wav wrote, bit for bit; so does the host-side
    conversion the harness uses when the vocoder has no device conversion"""
    from dict_tts_amd import infer
    from oracle import audio_ref
    g = np.load(os.path.join(golden_dir, "g9_save_wav.npz"))
    lens, wavs = gc.g9_wavs()
    for i, w in enumerate(wavs):
        for norm in (0, 1):
            want = g[f"u{i}.norm{norm}"]
            assert want.dtype == np.int16 and want.shape == (lens[i] * 256,)
            assert np.array_equal(audio_ref.save_wav_pcm(w, bool(norm)), want)
            assert np.array_equal(infer.wav_to_int16(w, bool(norm)), want)
