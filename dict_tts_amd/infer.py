"""Inference harness: the MI355X counterpart of ``DictTTSTask.test_step / after_infer / test_end``
(tasks/tts/dict_tts.py:179-311, tasks/tts/tts_base.py:329-376), batched.

Per batch: model.forward(infer=True) -> mel_out, pron_attn; ONE batched vocoder call (the reference runs the vocoder
once per utterance, and a second time on the ground-truth mel); per utterance: the waveform cut to its own length,
int16 scaling ``wav * 32767`` (utils/audio.py:11-16), the pinyin string decoded from ``pron_attn``
(dict_tts.py:294-304) and one ``meta.csv`` row with the reference's columns (dict_tts.py:305-311).
Plots, f0 and the ground-truth vocoder pass are not produced (they are not part of the hot path)."""
import csv
import os

import numpy as np
import torch
from scipy.io import wavfile

from .model import decode_pinyin_ids

META_COLUMNS = ["item_name", "text", "pinyin_tokens", "wav_fn_pred", "wav_fn_gt"]


def base_filename(results_id, item_name, text):
    """tasks/tts/dict_tts.py:257-261"""
    base_fn = f'[{results_id:06d}][{item_name.replace("%", "_")}][%s]'
    if text is not None:
        base_fn += text.replace(":", "$3A")[:80]
    return base_fn.replace(" ", "_")


def wav_to_int16(wav, norm=False):
    """utils/audio.py:11-16 (truncating cast, as numpy's astype does)"""
    wav = np.array(wav, dtype=np.float32, copy=True)
    if norm:
        wav = wav / np.abs(wav).max()
    wav *= 32767
    return wav.astype(np.int16)


def infer_batch(model, vocoder, batch, z_p=None):
    """batch: dict of tensors as DictTTSDataset.collater produces (word_tokens, keys, values, key_map, pinyin,
    pinyin_map, pron_modified).  Returns (outputs dict, list of float32 waveforms, one per utterance)."""
    out = _model_forward(model, batch, z_p)
    lens = out["mel_lens"]
    wav = vocoder.forward_batch(out["mel_out"], lens, check=True) if hasattr(vocoder, "overflowed") else vocoder.forward_batch(out["mel_out"], lens)
    hop = vocoder.hop
    lens_h = lens.cpu().tolist()
    wav_h = wav.cpu().numpy()
    return out, [wav_h[i, : lens_h[i] * hop] for i in range(len(lens_h))]


def _model_forward(model, batch, z_p):
    return model((batch["word_tokens"], batch.get("txt_tokens")), batch.get("pron_modified"), (None, None, None),
                 batch.get("ph2word"), None,
                 (batch["keys"], batch["values"], batch["key_map"], batch["pinyin"], batch["pinyin_map"]), infer=True, z_p=z_p)


def _iter_results(model, vocoder, batches, pipeline, out_wav_norm=False):
    """yields (batch, outputs, waveforms) per batch.  pipeline=True: the vocoder of batch i runs on a second HIP stream
    while text->mel of batch i+1 runs on the current one (the acoustic model's small kernels leave most CUs idle;
    bench.py measures this arrangement), and batch i's device->host copies happen after batch i+1 has been enqueued."""
    if not pipeline:
        for batch in batches:
            out, wavs = infer_batch(model, vocoder, batch, batch.get("z_p"))
            yield batch, out, wavs
        return
    main = torch.cuda.current_stream()
    voc_stream = torch.cuda.Stream()
    int16_on_device = hasattr(vocoder, "to_int16")
    hop = vocoder.hop

    st = {"redo_next": False}

    def finish(p):
        batch, out, wav, done = p
        done.synchronize()
        if hasattr(vocoder, "overflowed"):
            # the always-on detector (dict_tts_amd/vocoder.py): an fp16 operand overflowed in this batch — or in the previous one, in
            # which case this batch was already in flight in fp16 and is redone too.  check=True redoes in DTTS_VOC_BF16X3 (AUTO) or raises.
            bad = vocoder.overflowed()
            need, st["redo_next"] = bad or st["redo_next"], bad
            if need:
                with torch.cuda.stream(voc_stream):
                    wav = vocoder.forward_batch(out["mel_out"], out["mel_lens"], check=True)
                    if int16_on_device:
                        wav = vocoder.to_int16(wav, out["mel_lens"], norm=out_wav_norm)
        with torch.cuda.stream(voc_stream):
            wav_h = wav.cpu().numpy()          # int16 when the vocoder converts on the device (half the copy)
        lens_h = out["mel_lens"].cpu().tolist()
        return batch, out, [wav_h[i, : lens_h[i] * hop] for i in range(len(lens_h))]

    pending = None
    for batch in batches:
        out = _model_forward(model, batch, batch.get("z_p"))
        voc_stream.wait_stream(main)
        with torch.cuda.stream(voc_stream):
            wav = vocoder.forward_batch(out["mel_out"], out["mel_lens"])
            if int16_on_device:
                wav = vocoder.to_int16(wav, out["mel_lens"], norm=out_wav_norm)
        out["mel_out"].record_stream(voc_stream)
        out["mel_lens"].record_stream(voc_stream)
        done = torch.cuda.Event()
        done.record(voc_stream)
        if pending is not None:
            yield finish(pending)
        pending = (batch, out, wav, done)
    if pending is not None:
        yield finish(pending)


def run_inference(model, vocoder, batches, gen_dir, pinyin_encoder, sample_rate=22050, save_wavs=True, out_wav_norm=False,
                  pipeline=None):
    """batches: iterable of dicts with the tensors above plus 'item_name' (list[str]) and 'text' (list[str]).
    pinyin_encoder: list mapping pinyin-token id -> string (``pinyin_encoder.pkl`` of the reference).
    Writes <gen_dir>/wavs/*.wav and <gen_dir>/meta.csv; returns the meta rows.
    pipeline: overlap the vocoder with the next batch's text->mel on two streams (default: on for the HIP model)."""
    os.makedirs(os.path.join(gen_dir, "wavs"), exist_ok=True)
    rows, results_id = [], 0
    if pipeline is None:
        pipeline = hasattr(model, "ctx") and hasattr(vocoder, "ctx")
    for batch, out, wavs in _iter_results(model, vocoder, batches, pipeline, out_wav_norm):
        pron_attn = out["pron_attn"].cpu()
        for i, wav in enumerate(wavs):
            item_name, text = batch["item_name"][i], batch["text"][i]
            n_words = int((batch["word_tokens"][i] > 0).sum())
            ids = decode_pinyin_ids(pron_attn[i, :n_words], torch.as_tensor(batch["pinyin"][i][:n_words]))
            base_fn = base_filename(results_id, item_name, text)
            if save_wavs:
                pcm = wav if wav.dtype == np.int16 else wav_to_int16(wav, out_wav_norm)   # int16: converted on the device
                wavfile.write(os.path.join(gen_dir, "wavs", (base_fn % "P") + ".wav"), sample_rate, pcm)
            rows.append({"item_name": item_name, "text": text.replace(",", "，").replace(".", "。"),
                         "pinyin_tokens": " ".join(pinyin_encoder[j] for j in ids),
                         "wav_fn_pred": base_fn % "P", "wav_fn_gt": base_fn % "G"})
            results_id += 1
    with open(os.path.join(gen_dir, "meta.csv"), "w", newline="", encoding="utf-8") as f:
        w = csv.writer(f)
        w.writerow([""] + META_COLUMNS)      # pandas' DataFrame.to_csv layout: leading index column
        for k, r in enumerate(rows):
            w.writerow([k] + [r[c] for c in META_COLUMNS])
    return rows
