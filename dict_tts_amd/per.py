"""Pronunciation error rate of a generated ``meta.csv`` against the labelled test set: the counterpart of
``scripts/get_pron_error.py`` (47 lines) without its jiwer / pypinyin dependencies.

* gold (``scripts/pron_label/label_set0.csv``, get_pron_error.py:9-18): column ``ph`` is
  ``<BOS> k a3 | er3 | p u3 # p ei2 ... <EOS>``; the 6 characters at either end are dropped, syllables are split on
  `` | `` / `` # `` and each syllable's phones are glued together (``ka3 er3 pu3 pei2 ...``).
* pred (``meta.csv`` written by infer.run_inference, get_pron_error.py:31-44): column ``pinyin_tokens`` holds
  two tokens (initial, final) per character; ``<UNK> `` is removed and consecutive pairs are glued.
* PER = 100 * word error rate over the whole set with the PREDICTION as the first (truth) argument, exactly as the
  reference calls ``wer(pred, gold)`` (get_pron_error.py:47): (S + D + I) summed over sentences / words in pred.
The heteronym count of the reference (pypinyin) is not reproduced.
"""
import re
import sys


def gold_from_label_csv(path):
    gold, word_num = [], 0
    with open(path, "r", encoding="utf-8") as f:
        lines = f.readlines()
    for line in lines[1:]:
        pron_label = line.split(",")[4]
        prons = [item.replace(" ", "") for item in re.split(r" \| | \# ", pron_label[6:-6])]
        word_num += len(prons)
        gold.append(" ".join(prons))
    return gold, word_num


def pred_from_meta_csv(path):
    pred = []
    with open(path, "r", encoding="utf-8") as f:
        lines = f.readlines()
    for line in lines[1:]:
        toks = line.split(",")[3].replace("<UNK> ", "").replace("\n", "").split(" ")
        pred.append(" ".join(toks[i] + toks[i + 1] for i in range(0, len(toks) - 1, 2)))
    return pred


def edit_distance(a, b):
    """Levenshtein distance between two token lists (substitution, deletion, insertion all cost 1)"""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[-1]


def wer(truth, hypothesis):
    """corpus-level word error rate of two equally long lists of space-separated sentences"""
    if len(truth) != len(hypothesis):
        raise ValueError(f"{len(truth)} truth sentences vs {len(hypothesis)} hypotheses")
    edits = words = 0
    for t, h in zip(truth, hypothesis):
        tw, hw = t.split(), h.split()
        edits += edit_distance(tw, hw)
        words += len(tw)
    if words == 0:
        raise ValueError("no words in the truth sentences")
    return edits / words


def pron_error_rate(meta_csv, label_csv):
    gold, word_num = gold_from_label_csv(label_csv)
    pred = pred_from_meta_csv(meta_csv)
    return 100.0 * wer(pred, gold), word_num


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit("usage: python -m dict_tts_amd.per <generated/meta.csv> <label_set0.csv>")
    per, n = pron_error_rate(sys.argv[1], sys.argv[2])
    print(f"Word num: {n}")
    print("PER: ", "%.2f" % per, "%")
