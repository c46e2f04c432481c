#!/usr/bin/env python3
"""Derive the *structure* of the Biaobei test sentences (SURVEY.md §8d "Config 2") as a small data fixture.

TEST/BENCH INFRASTRUCTURE — runs only in the build container, where /root/reference exists.

Reads two reference DATA files (no reference code is imported or copied):
  * scripts/pron_label/label_set0.csv  — the 200 Biaobei test sentences (column ``txt``)
  * data/zh-dict.json                  — char -> {pinyin: [gloss, ...]}
and writes ``dict_tts_amd/data/biaobei_struct.json`` holding only integers:
  sentences : list of lists of word ids (3 + rank of the char in the sorted distinct-char set,
              the id rule of utils/text_encoder.py:5-13,197-205)
  entries   : word id -> list of senses, each [gloss_tokens, pinyin_initial_id, pinyin_final_id]
              following data_gen/tts/binarizer_zh.py:261-307: one sense per pronunciation, gloss length
              min(len(gloss), 30) + 2 ([CLS]/[SEP]); chars absent from the dict get the single
              3-token zero entry with pinyin '<UNK>' (binarizer_zh.py:250-259).
The gloss *embeddings* cannot be produced offline (roformer weights absent); they are
generated from a counter-based RNG in dict_tts_amd/synth.py.
"""
import csv
import json
import os
import re
import sys
import unicodedata

REF = os.environ.get("DICT_TTS_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dict_tts_amd", "data", "biaobei_struct.json")

INITIALS = ["zh", "ch", "sh", "b", "p", "m", "f", "d", "t", "n", "l", "g", "k", "h", "j", "q", "x", "r", "z", "c",
            "s", "y", "w"]
TONE_MARKS = {"̄": 1, "́": 2, "̌": 3, "̀": 4}


def split_pinyin(p):
    """own initial / final+tone splitter (pypinyin is not installed); returns two token strings."""
    dec = unicodedata.normalize("NFD", p)
    tone = 5
    base = ""
    for ch in dec:
        if ch in TONE_MARKS:
            tone = TONE_MARKS[ch]
        elif not unicodedata.combining(ch):
            base += ch
    base = base.replace("ü", "v").lower()
    ini = ""
    for cand in INITIALS:
        if base.startswith(cand):
            ini = cand
            break
    fin = base[len(ini):] + str(tone)
    return ini, fin


def write_full_dictionary(zh):
    """BASELINE.json configs[4] ("word_size=8000, full zh-dict.json entry set in HBM"): the sense structure of ALL
    zh-dict.json characters, same per-sense rule as above.  Word id = 3 + rank of the character in the sorted key set
    (ids 3 .. 3 + n - 1 < word_size 8000); pinyin-token ids in first-seen order after '<UNK>', folded into 1..184
    (value_embedding_size = 185).  -> dict_tts_amd/data/zh_dict_struct.json (integers only)."""
    chars = sorted(zh)
    pinyin_tokens = ["<UNK>"]

    def pid(tok):
        if tok not in pinyin_tokens:
            pinyin_tokens.append(tok)
        i = pinyin_tokens.index(tok)
        return i if i <= 184 else 1 + (i % 184)

    entries = {}
    for r, c in enumerate(chars):
        senses = []
        for pinyin, glosses in zh[c].items():
            gloss = "".join(glosses).replace("～", c)
            gloss = re.sub(r"[^一-鿿，。！？；：、,.!?;:]", "", gloss)
            ini, fin = split_pinyin(pinyin)
            senses.append([min(len(gloss), 30) + 2, pid(ini), pid(fin)])
        entries[3 + r] = senses
    out = {"source": "zh-dict.json, all characters; integers only", "n_entries": len(chars),
           "entries": {str(k): v for k, v in entries.items()}}
    path = os.path.join(os.path.dirname(OUT), "zh_dict_struct.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    ns = [len(v) for v in entries.values()]
    print("full dictionary:", len(chars), "entries, senses max", max(ns), "heteronyms", sum(n > 1 for n in ns),
          "gloss tokens", sum(s[0] for v in entries.values() for s in v), "pinyin tokens", len(pinyin_tokens), "bytes",
          os.path.getsize(path))


def main():
    zh = json.load(open(os.path.join(REF, "data", "zh-dict.json"), encoding="utf-8"))
    rows = list(csv.DictReader(open(os.path.join(REF, "scripts", "pron_label", "label_set0.csv"), encoding="utf-8")))
    texts = []
    for r in rows:
        t = re.sub(r"[^一-鿿]", "", r["txt"])  # characters only; punctuation carries no dict entry
        texts.append(t)
    chars = sorted(set("".join(texts)))
    cid = {c: 3 + i for i, c in enumerate(chars)}
    pinyin_tokens = ["<UNK>"]

    def pid(tok):
        if tok not in pinyin_tokens:
            pinyin_tokens.append(tok)
        return pinyin_tokens.index(tok)

    entries = {}
    for c in chars:
        if c not in zh:
            entries[cid[c]] = [[3, 0, -1]]  # zero entry: 3 gloss tokens, single '<UNK>' pinyin token
            continue
        senses = []
        for pinyin, glosses in zh[c].items():
            gloss = "".join(glosses).replace("～", c)
            gloss = re.sub(r"[^一-鿿，。！？；：、,.!?;:]", "", gloss)
            n = min(len(gloss), 30) + 2
            ini, fin = split_pinyin(pinyin)
            senses.append([n, pid(ini), pid(fin)])
        entries[cid[c]] = senses[:6]
    n_tok = len(pinyin_tokens)
    if n_tok > 184:  # value_embedding_size = 185 (egs/egs_bases/tts/dict_tts.yaml); fold the overflow
        for senses in entries.values():
            for s in senses:
                for j in (1, 2):
                    if s[j] > 184:
                        s[j] = 1 + (s[j] % 184)
    out = {
        "source": "label_set0.csv (200 Biaobei test rows) x zh-dict.json; integers only",
        "n_pinyin_tokens": min(n_tok, 185),
        "sentences": [[cid[c] for c in t] for t in texts],
        "entries": {str(k): v for k, v in sorted(entries.items())},
    }
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    write_full_dictionary(zh)
    lens = [len(s) for s in out["sentences"]]
    print("sentences", len(lens), "chars/sent min/mean/max", min(lens), sum(lens) / len(lens), max(lens),
          "distinct chars", len(chars), "pinyin tokens", n_tok, "bytes", os.path.getsize(OUT))


if __name__ == "__main__":
    sys.exit(main())
