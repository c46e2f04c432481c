"""CPU test of the inference harness (dict_tts_amd/infer.py) with stand-in model / vocoder objects: file naming,
int16 scaling, pinyin decode per utterance length, meta.csv layout (tasks/tts/dict_tts.py:257-311)."""
import csv
import os

import numpy as np
import torch
from scipy.io import wavfile

from dict_tts_amd import infer, synth


class _FakeModel:
    def __call__(self, txt_tokens, pron_modified, kvm, ph2word, word_len, dict_msg, infer=False, z_p=None):
        wt = txt_tokens[0]
        B, Tw = wt.shape
        P = dict_msg[3].shape[2]
        pa = torch.zeros(B, Tw, P)
        pa[:, :, 0] = 0.4
        if P > 2:
            pa[:, 2, 2] = 0.9          # word 2 picks its second sense (tokens 2,3) when it has one
        lens = torch.tensor([8 + 4 * b for b in range(B)], dtype=torch.int32)
        return {"mel_out": torch.zeros(B, int(lens.max()), 80), "pron_attn": pa, "mel_lens": lens}


class _FakeVocoder:
    hop = 256

    def forward_batch(self, mel, lens):
        B, T, _ = mel.shape
        t = torch.arange(T * self.hop, dtype=torch.float32)
        return (0.5 * torch.sin(t / 20.0)).repeat(B, 1)


def test_run_inference_outputs(tmp_path):
    st = synth.biaobei_struct()
    b = synth.make_batch(st["sentences"][:2], 1234)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    batch["item_name"] = ["000001", "a%b"]
    batch["text"] = ["卡尔普陪外孙玩滑梯.", "x: y, z"]
    enc = [f"p{i}" for i in range(185)]
    rows = infer.run_inference(_FakeModel(), _FakeVocoder(), [batch], str(tmp_path), enc)
    assert [r["item_name"] for r in rows] == ["000001", "a%b"]
    assert rows[0]["wav_fn_pred"] == "[000000][000001][P]卡尔普陪外孙玩滑梯." and rows[0]["wav_fn_gt"].endswith("[G]卡尔普陪外孙玩滑梯.")
    assert rows[1]["wav_fn_pred"] == "[000001][a_b][P]x$3A_y,_z" and rows[1]["text"] == "x: y， z"
    # two pinyin tokens per inner word, only over this utterance's own words
    n0 = int((b["word_tokens"][0] > 0).sum())
    assert len(rows[0]["pinyin_tokens"].split(" ")) == 2 * (n0 - 2)
    sr, w = wavfile.read(os.path.join(tmp_path, "wavs", rows[0]["wav_fn_pred"] + ".wav"))
    assert sr == 22050 and w.dtype == np.int16 and w.shape == (8 * 256,)
    assert w[31] == np.int16(np.float32(0.5 * np.sin(np.float32(31) / np.float32(20.0))) * np.float32(32767))
    sr, w1 = wavfile.read(os.path.join(tmp_path, "wavs", rows[1]["wav_fn_pred"] + ".wav"))
    assert w1.shape == (12 * 256,)
    got = list(csv.reader(open(os.path.join(tmp_path, "meta.csv"), encoding="utf-8")))
    assert got[0] == ["", "item_name", "text", "pinyin_tokens", "wav_fn_pred", "wav_fn_gt"] and got[1][0] == "0" and len(got) == 3


def test_int16_scaling_matches_reference_rule():
    x = np.array([0.0, 0.5, -0.5, 0.99997, -1.0], np.float32)
    assert infer.wav_to_int16(x).tolist() == [0, 16383, -16383, 32766, -32767]
    assert infer.wav_to_int16(x, norm=True).tolist() == [0, 16383, -16383, 32766, -32767]


def test_pron_error_rate_counterpart(tmp_path):
    """dict_tts_amd.per: gold syllables from the label csv's ph column, predicted syllables from meta.csv's
    pinyin_tokens pairs, PER = WER(pred as truth, gold) * 100 (scripts/get_pron_error.py:9-18,31-47)"""
    from dict_tts_amd import per
    label = tmp_path / "label.csv"
    label.write_text(",item_name,spk,txt,ph,wav_fn,others\n"
                     "0,1,SPK1,卡尔普.,<BOS> k a3 | er3 | p u3 <EOS>,a.wav,{}\n"
                     "1,2,SPK1,别再.,<BOS> b ie2 # z ai4 <EOS>,b.wav,{}\n", encoding="utf-8")
    meta = tmp_path / "meta.csv"
    meta.write_text(",item_name,text,pinyin_tokens,wav_fn_pred,wav_fn_gt\n"
                    "0,1,卡尔普。,k a3 <UNK> er3 p u4,x,y\n"       # '<UNK> ' is dropped first, then tokens pair up by position
                    "1,2,别再。,b ie2 z ai4,x,y\n", encoding="utf-8")
    gold, n = per.gold_from_label_csv(str(label))
    assert gold == ["ka3 er3 pu3", "bie2 zai4"] and n == 5
    pred = per.pred_from_meta_csv(str(meta))
    assert pred == ["ka3 er3p", "bie2 zai4"]                         # pairs are positional, as in the reference
    assert per.edit_distance("a b c".split(), "a x c d".split()) == 2
    assert per.wer(["a b c", "d"], ["a b c", "e"]) == 0.25
    rate, _ = per.pron_error_rate(str(meta), str(label))
    # pred as truth: "ka3 er3p" vs gold "ka3 er3 pu3" -> 1 substitution + 1 insertion over 4 predicted words
    assert abs(rate - 100.0 * 2 / 4) < 1e-9
