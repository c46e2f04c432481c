"""Reader of the reference's ``dict_embed`` indexed dataset and its conversion to the resident dictionary table.

The reference looks every character of every sentence up in ``<binary_data_dir>/dict_embed.{idx,data}``
(tasks/tts/dataset_utils.py:305-330): an ``IndexedDataset`` (utils/indexed_datasets.py:7-54 — ``.idx`` is a numpy-saved
dict ``{'offsets': [...]}`` of byte offsets, ``.data`` the concatenated pickles) whose item ``i`` belongs to word id
``i`` and is the dict written by the binarizer (data_gen/tts/binarizer_zh.py:250-259,301-309):

    tokens_gloss: list[str] · key: f32 tensor [L,768] · value: f32 tensor [L,768] · key_map: list[int] (len L)
    pinyin: list[str] (len P, strings of ``pinyin_encoder.pkl``) · pinyin_map: list[int] (len P)

``table_from_dict_embed`` turns the whole dataset into the ragged arrays ``dtts_dict_table_upload`` takes
(include/dicttts_hip.h), once; afterwards a batch is just the word ids (``make_id_batch``).
"""
import pickle

import numpy as np


def read_indexed_dataset(path):
    """-> list of items of ``<path>.idx`` / ``<path>.data`` (utils/indexed_datasets.py:7-39, without the cache)"""
    offsets = np.load(f"{path}.idx", allow_pickle=True).item()["offsets"]
    items = []
    with open(f"{path}.data", "rb") as f:
        for i in range(len(offsets) - 1):
            f.seek(offsets[i])
            items.append(pickle.loads(f.read(offsets[i + 1] - offsets[i])))
    return items


def _np(t, dtype):
    return np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t), dtype=dtype)


def table_from_items(items, pinyin_encoder):
    """items: the ``dict_embed`` items in word-id order; pinyin_encoder: list of pinyin strings (``pinyin_encoder.pkl``).
    -> dict(tok_off i32 [n+1], keys f32 [sum_L,D], values f32 [sum_L,D] or None when every value equals its key,
            key_map f32 [sum_L], pin_off i32 [n+1], pinyin i64 [sum_P], pinyin_map i64 [sum_P], L, P, ids)"""
    index = {tok: i for i, tok in reversed(list(enumerate(pinyin_encoder)))}   # list.index(): the first occurrence
    n = len(items)
    tok_off, pin_off = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    keys, values, key_map, pinyin, pinyin_map = [], [], [], [], []
    same = True
    for i, it in enumerate(items):
        k, v = _np(it["key"], np.float32), _np(it["value"], np.float32)
        km = np.asarray(it["key_map"], np.float32)
        pm = np.asarray(it["pinyin_map"], np.int64)
        py = np.array([index[t] for t in it["pinyin"]], np.int64)   # KeyError = a token the pinyin encoder does not hold
        if k.ndim != 2 or k.shape != v.shape or km.shape != (k.shape[0],) or pm.shape != py.shape:
            raise ValueError(f"dict_embed item {i}: inconsistent shapes key {k.shape} value {v.shape} key_map {km.shape} "
                             f"pinyin {py.shape} pinyin_map {pm.shape}")
        same = same and (it["key"] is it["value"] or np.array_equal(k, v))
        tok_off[i + 1] = tok_off[i] + k.shape[0]
        pin_off[i + 1] = pin_off[i] + py.shape[0]
        keys.append(k)
        values.append(v)
        key_map.append(km)
        pinyin.append(py)
        pinyin_map.append(pm)
    return {"ids": {i: i for i in range(n)}, "tok_off": tok_off, "pin_off": pin_off, "keys": np.concatenate(keys),
            "values": None if same else np.concatenate(values), "key_map": np.concatenate(key_map),
            "pinyin": np.concatenate(pinyin), "pinyin_map": np.concatenate(pinyin_map),
            "L": np.diff(tok_off), "P": np.diff(pin_off)}


def table_from_dict_embed(path, pinyin_encoder):
    """path: ``<binary_data_dir>/dict_embed`` (without extension); pinyin_encoder: list, or the path of pinyin_encoder.pkl"""
    if isinstance(pinyin_encoder, str):
        with open(pinyin_encoder, "rb") as f:
            pinyin_encoder = pickle.load(f)
    return table_from_items(read_indexed_dataset(path), pinyin_encoder)


def entry_ids_for_words(words, token_to_id, n_entries):
    """the reference's per-character lookup (dataset_utils.py:313-318): unknown words use entry 2 ('<UNK>')"""
    out = []
    for w in words:
        i = token_to_id.get(w, 2)
        if not 0 <= i < n_entries:
            raise IndexError(f"word {w!r} has id {i}, the table holds {n_entries} entries")
        out.append(i)
    return np.array(out, np.int32)
