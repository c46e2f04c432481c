#!/usr/bin/env python3
"""Headline benchmark of the Dict-TTS inference hot path on MI355X (BASELINE.json: mel-frames/sec + audio-samples/sec
(RTF) per GPU, Biaobei batch=60).

One "step" = one pass of the whole hot path over one batch of 60 synthetic Biaobei utterances per GPU, as a deployment
runs it (the dictionary is resident in HBM, uploaded once before the timed region — SURVEY.md 8f-1):

    H2D of the batch's ids (word tokens, dictionary entry ids, forced senses: a few KB)
    -> S2PA dictionary encoder -> duration predictor -> length regulator (the one T_mel host sync)
    -> prior sample drawn on the device -> prior flow + FVAE decoder -> HifiGAN (DTTS_VOC_F16, the waveform-exact mode)
    -> int16 conversion on the device -> D2H of the int16 waveforms into pinned host memory

The timed loop rotates over 4 DIFFERENT batches of the 200-sentence test set (different lengths, T_w, L_k, T_mel), and the
H2D / D2H copies are inside the timed region.  Two labelled side figures are measured after it in the same process:
"inputs_resident" (ids already in HBM, fp32 waveform left in HBM: round 1's definition) and "tensor_api_incl_h2d" (the
reference's own API: keys / values tensors [B,T_w,L_k,768] uploaded per batch, 1.47 GB at B=60).
Data: synthetic (counter-based weights / gloss embeddings, real Biaobei sentence + dictionary structure); durations
are PREDICTED by the duration predictor (its output bias is set so that the mean is ~22 frames per character).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  --workload rotating (default): weak scaling, every rank rotates over its own 4 batches
of 60; with N > 1 the mels are all-gathered over RCCL after a (B, T_mel) exchange (dict_tts_amd/shard.py).
--workload testset: BASELINE configs[2] — a step is ONE pass over all 200 test sentences, utterance i -> rank i mod N in
batches of <= 60 (tasks/tts/tts_base.py:114-127,148-151); strong scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FRAME_VOCODER = 614_105_088   # 2*MAC of HifiGanGenerator per mel frame
FLOP_PER_FRAME_DECODER = 4_691_968     # A8-A10 per (padded) mel frame
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)
DATA_CEILING_F16_TFLOPS = 1670.0       # what a register-resident v_mfma_f32_32x32x16_f16 loop sustains on this chip with NON-ZERO fp16 operands, two waves
                                       # per SIMD (tools/micro/mfma_peak: 1,669.8; 2,342-2,485 with all-zero operands; profiles/r05_*_mfma_ceiling.txt)
PEAK_HBM_GBS = 8000.0
DUR_BIAS = 3.09                        # exp(softplus(3.09)) - 1 ~= 22 frames per word
N_TEST = 200                           # Biaobei test rows (label_set0.csv)
C5_BATCH = 128                         # BASELINE.json configs[4]: batch=128 mixed-length (over 4 GPUs there)


# ---------------------------------------------------------------------------------------------------------------- CPU leg
def host_cpu():
    """(model string, physical cores available to this process, logical CPUs available)"""
    aff = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    model, phys, cur = "unknown", set(), {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if ":" in line:
                    k, v = [x.strip() for x in line.split(":", 1)]
                    cur[k] = v
                elif not line.strip():
                    if cur and int(cur.get("processor", -1)) in aff:
                        phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                        model = cur.get("model name", model)
                    cur = {}
        if cur and int(cur.get("processor", -1)) in aff:
            phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
            model = cur.get("model name", model)
    except OSError:
        pass
    return model, (len(phys) or len(aff)), len(aff)


def cpu_baseline(torch, np, synth, voc, mode="protocol", budget_s=420.0):
    """The CPU oracle (our restatement of the reference, oracle/*.py, pinned by tests/golden) timed on the host cores per
    BASELINE.md §3: reference-faithful protocol = B=1 per utterance, model forward + one spec2wav (tasks/tts/dict_tts.py:179-255),
    3 warm-ups, median of 5 repetitions over ALL 60 rows of the 60-utterance set; plus a batched variant.
      thread sweep: rows 0..5 x 2 passes at {8, 16, 32, 64, physical cores} torch threads (those that the box has); the best count runs the
                    protocol and is stated as `cores`, all three rates are listed (a B=1 forward does not scale to 128 threads);
      mode 'protocol' (default): all 60 rows, up to 5 repetitions inside a time budget (`budget_s`, at least one full repetition;
                    the number completed is stated), batched variant B=60 (one pass at the best thread count of a B=8 sweep);
      mode 'full':  no budget, 5 repetitions, batched variant B=60 (3 timed passes after a warm-up);
      mode 'bounded': rows 0..2 x 5 repetitions, batched B=4 (round 2's default, ~35 s).
    The same leg checks the GPU vocoder against the oracle waveform on the first utterance's ORACLE mel (this is the only place
    bench.py may use the oracle)."""
    from oracle import dict_tts_ref as ref
    from oracle import hifigan_ref as href
    model_name, phys, logical = host_cpu()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd_np = synth.dict_tts_state_dict(1234)
    sd_np["dur_predictor.linear.0.bias"] = np.array([DUR_BIAS], np.float32)
    sd = href.fold_weight_norm({k: T(v) for k, v in sd_np.items()})
    hsd = href.fold_weight_norm({k: T(v) for k, v in synth.hifigan_state_dict(1234).items()})
    cfg = synth.hifigan_config()
    st = synth.biaobei_struct()
    n_utt = 3 if mode == "bounded" else 60
    n_batched = {"bounded": 4, "protocol": 60, "full": 60}[mode]
    rms = lambda a: float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64)))))
    inputs = {}

    def one(i, keep=False):
        if i not in inputs:   # (input construction is not part of the reference's timed region either: the DataLoader workers build it)
            inputs[i] = ({k: T(v) for k, v in synth.make_batch([st["sentences"][i]], 1234).items()}, T(synth.noise(1234, 1, 1024, f"cpu.z{i}")))
        b, z = inputs[i]
        t0 = time.perf_counter()
        r = ref.forward_infer(sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                              b["pron_modified"], z_p=lambda B, T4: z[:, :, :T4])
        t1 = time.perf_counter()
        wav = href.spec2wav(hsd, cfg, r["mel_out"][0].numpy())
        t2 = time.perf_counter()
        assert wav.numel() == r["mel_out"].shape[1] * 256
        return int(r["mel_out"].shape[1]), t1 - t0, t2 - t1, ((r["mel_out"][0].numpy(), wav.numpy()) if keep else None)

    t_start = time.perf_counter()
    # ---- thread sweep (the first pass of each setting is its warm-up)
    sweep = {}
    cands = sorted({t for t in (8, 16, 32, 64, phys) if 1 <= t <= max(1, logical)}) or [max(1, phys)]
    if mode == "bounded":
        cands = [max(1, phys)]
    for th in cands:
        torch.set_num_threads(th)
        best = 0.0
        for rep in range(2 if len(cands) > 1 else 1):
            fr = tt = 0.0
            for i in range(min(6, n_utt)):
                f, a, v, _ = one(i)
                fr, tt = fr + f, tt + a + v
            if rep or len(cands) == 1:
                best = fr / tt
        sweep[str(th)] = best
    threads = int(max(sweep, key=lambda k: sweep[k])) if len(cands) > 1 else cands[0]
    torch.set_num_threads(threads)
    kept_all = []
    for i in range(3):                       # 3 warm-ups (rows 0-2, untimed) at the chosen thread count; they double as the waveform checks
        f, _, _, k = one(i, keep=True)
        kept_all.append(k)
    reps = []
    for r_i in range(5):
        fr = t_a = t_v = 0.0
        for i in range(n_utt):
            f, a, v, _ = one(i)
            fr, t_a, t_v = fr + f, t_a + a, t_v + v
        reps.append((fr / (t_a + t_v), fr / t_a, fr / t_v, fr))
        if mode == "protocol" and time.perf_counter() - t_start > budget_s * (r_i + 1) / (r_i + 2):
            break                            # the next repetition would not fit the budget
    n_reps = len(reps)
    reps.sort()
    e2e, t2m, vocr, frames = reps[len(reps) // 2]
    # batched variant: the first n_batched rows as ONE batch (text->mel) + one batched generator call on the padded mel.
    # A B = 60 forward scales to more threads than a B = 1 one: its thread count is the best of {B=1's best, 32, 64, physical} measured on a
    # B = 8 batch (listed), then the full batch runs once at that count (mode 'full': 3 timed passes after a warm-up).
    def batched_pass(nb, zb_):
        bb_ = batched_inputs[nb]
        t0 = time.perf_counter()
        r = ref.forward_infer(sd, bb_["word_tokens"], (bb_["keys"], bb_["values"], bb_["key_map"], bb_["pinyin"], bb_["pinyin_map"]),
                              bb_["pron_modified"], z_p=lambda B, T4: zb_[:B, :, :T4])
        with torch.no_grad():
            href.generator_forward(hsd, cfg, r["mel_out"].transpose(1, 2).contiguous())
        return int((r["mel2word"] > 0).sum()) / (time.perf_counter() - t0)

    zb = T(synth.noise(1234, n_batched, 1024, "cpu.zb"))
    batched_inputs = {nb: {k: T(v) for k, v in synth.make_batch(st["sentences"][:nb], 1234).items()} for nb in sorted({min(8, n_batched), n_batched})}
    b_sweep = {}
    b_threads = threads
    if mode != "bounded":
        for th in sorted({t for t in (threads, 32, 64, phys) if 1 <= t <= max(1, logical)}):
            torch.set_num_threads(th)
            b_sweep[str(th)] = batched_pass(min(8, n_batched), zb)
        b_threads = int(max(b_sweep, key=lambda k: b_sweep[k]))
    torch.set_num_threads(b_threads)
    bt = []
    for rep in range(4 if mode == "full" else 1):          # full: the first pass is a warm-up; otherwise one pass, the threads are warm
        v = batched_pass(n_batched, zb)
        if rep or mode != "full":
            bt.append(v)
    bt.sort()
    torch.set_num_threads(threads)
    out = {"value": e2e, "unit": "mel-frames/s", "cores": threads, "kind": "port",
           "cpu_model": model_name, "physical_cores": phys, "logical_cpus": logical,
           "thread_sweep_mel_frames_per_s": sweep,
           "protocol": "BASELINE.md §3: B=1 per utterance (text->mel + one spec2wav), 3 warm-ups, median of the repetitions; "
                       "torch CPU fp32 oracle (oracle/dict_tts_ref.py + oracle/hifigan_ref.py); threads = the best of the sweep "
                       "{8, 16, 32, 64, physical} measured on rows 0..5",
           "sample": f"rows 0..{n_utt - 1} of the 60-utterance set x {n_reps} repetitions ({int(frames)} frames per repetition), mode={mode}"
                     + (f", time budget {budget_s:.0f} s" if mode == "protocol" else ""),
           "repetitions": n_reps,
           "text2mel_frames_per_s": t2m, "vocoder_frames_per_s": vocr, "samples_per_s": e2e * 256, "rtf": 22050.0 / (e2e * 256),
           "batched": {"value": bt[len(bt) // 2], "unit": "valid mel-frames/s", "B": n_batched, "cores": b_threads,
                       "thread_sweep_on_B8_mel_frames_per_s": b_sweep,
                       "note": "one forward_infer + one generator call on the padded batch (padding frames are computed, not counted)"},
           "cpu_seconds": time.perf_counter() - t_start}
    if voc is not None:   # waveform gates of the benched vocoder mode, measured on this box
        out["waveform_check"] = waveform_check(np, voc, kept_all)
    out["_kept"] = kept_all   # (popped by the caller: the other vocoder mode is checked on the same utterances)
    return out


def waveform_check(np, voc, kept_all):
    """BASELINE.md §4 waveform gates, MEASURED: the oracle's mel of utterances 0..2 -> GPU vocoder (batched call, ragged lengths) against
    the oracle vocoder's waveform of each; worst case over the three."""
    rms = lambda a: float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64)))))
    ws = voc.spec2wav_batch([m for m, _ in kept_all])
    per = [{"frames": int(m.shape[0]), "rms_diff": rms(w - wr), "abs_rms_delta": abs(rms(w) - rms(wr))} for w, (m, wr) in zip(ws, kept_all)]
    worst = max(per, key=lambda d: d["rms_diff"])
    return {"vocoder_mode": voc_mode_name(voc), "frames": sum(d["frames"] for d in per), "utterances": per,
            "rms_diff": worst["rms_diff"], "abs_rms_delta": max(d["abs_rms_delta"] for d in per), "gate": 1e-4,
            "pass": bool(all(d["rms_diff"] <= 1e-4 and d["abs_rms_delta"] <= 1e-4 for d in per)),
            "input": "utterances 0..2: the ORACLE's mel -> GPU vocoder (one ragged batch) vs the oracle vocoder, worst of the three"}


def voc_mode_name(voc):
    from dict_tts_amd import abi
    return {v: k for k, v in abi.VOC_PRECISIONS.items()}[voc.precision]


# ---------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=60)
    ap.add_argument("--precision", choices=["f16", "bf16", "bf16x3"], default="f16",
                    help="vocoder arithmetic (include/dicttts_hip.h): f16 = the waveform-exact default")
    ap.add_argument("--workload", choices=["rotating", "testset", "config5"], default="rotating")
    ap.add_argument("--input", choices=["table", "resident", "tensors"], default="table",
                    help="what a timed step includes: table = ids H2D + int16 waveform D2H (default); resident = ids already in "
                         "HBM, fp32 waveform stays in HBM; tensors = the reference API, keys/values uploaded per batch")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of mels when --gpus > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["protocol", "bounded", "full"], default="protocol",
                    help="protocol (default): all 60 rows, B=1, up to 5 repetitions inside a time budget, thread sweep; bounded: rows 0..2 (~35 s); "
                         "full: no budget, B=60 batched variant")
    ap.add_argument("--cpu-budget", type=float, default=420.0, help="seconds the 'protocol' CPU leg may spend on its repetitions")
    ap.add_argument("--no-side", action="store_true", help="skip the side figures / per-stage / second-mode measurements")
    ap.add_argument("--voc-priority", type=int, default=0, help="HIP stream priority of the vocoder stream (two-stream pipeline): 0 normal, -1 high")
    ap.add_argument("--voc-tune", type=int, default=0, help="dtts_config.tune_flags of the vocoder context (A/B switches, include/dicttts_hip.h)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the vocoder on the text->mel stream (default: vocoder of batch i on a second HIP stream, "
                         "overlapping text->mel of batch i+1; every batch is still fully processed inside the timed region)")
    ap.add_argument("--lib", default=None, help="path of a library build to load instead of the in-tree release library (same-box A/B runs "
                                                "with the ablation build; the headline is always measured on the release library)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from dict_tts_amd import abi, model, synth, vocoder
    if args.lib:
        abi.load_library(os.path.abspath(args.lib))
    from dict_tts_amd.shard import exchange_shapes, gather_mels, n_steps, ranks_seen, shard_indices

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # test hook (1-GPU box): DTTS_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 and uses gloo, so that the N>1 control
    # flow (rendezvous, sharding, barriers, max-over-ranks timing, frame all-reduce, mel all-gather) runs without N GPUs
    one_dev = os.environ.get("DTTS_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    seen = None
    if dist is not None:   # (rank, device index, device uuid) of every rank through the collective backend: a real N-GPU run shows N distinct devices
        try:
            uuid = str(torch.cuda.get_device_properties(dev).uuid)
        except Exception:
            uuid = ""
        seen = ranks_seen(dist, device_index=local_rank, device_uuid=uuid)
        assert seen is not None and len(seen) == world and sorted(r["rank"] for r in seen) == list(range(world)), seen
        if not one_dev:   # a real multi-GPU run: the collective backend must have seen `world` DISTINCT devices
            assert len({(r["device_index"], r["device_uuid"]) for r in seen}) == world, f"ranks share a device: {seen}"

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    # ---- weights (random-init of the real architecture) and the acoustic model / vocoder behind the reference APIs
    sd_np = synth.dict_tts_state_dict(1234)
    sd_np["dur_predictor.linear.0.bias"] = np.array([DUR_BIAS], np.float32)
    m = model.PortaSpeech_dict(hparams={})
    m.load_state_dict({k: T(v) for k, v in sd_np.items()})
    voc_sd = {k: T(v) for k, v in synth.hifigan_state_dict(1234).items()}
    voc = vocoder.HifiGAN(state_dict=voc_sd, config={**synth.hifigan_config(), "dtts_tune_flags": args.voc_tune},
                          precision=abi.VOC_PRECISIONS[args.precision])
    voc.ctx.timer_enable(abi.TIMER_VOC_CONV)
    hop = voc.hop
    # ---- the dictionary: resident in HBM, uploaded once (not part of a step)
    st = synth.biaobei_struct()
    c5_sents = None
    if args.workload == "config5":   # BASELINE configs[4]: the FULL zh-dict.json entry set resident (7,030 entries), B = 128 mixed-length utterances
        full = synth.zh_dict_struct()
        table = synth.dict_table(1234, full["entries"])
        c5_sents = synth.config5_sentences(C5_BATCH, 55, full["entries"])
    else:
        table = synth.dict_table(1234)
    m.upload_dict_table(table)
    live_rows_of_entry = np.add.reduceat((table["key_map"] != 0).astype(np.int64), table["tok_off"][:-1])

    # ---- this rank's batches (host side, pinned): sentence index lists
    if args.workload == "config5":   # ONE batch of 128 utterances per step, utterance i -> rank i mod N (4 GPUs x 32 in BASELINE.json)
        idx_lists = shard_indices(C5_BATCH, rank, world, (C5_BATCH + world - 1) // world)
        steps_per_pass = n_steps(C5_BATCH, world, (C5_BATCH + world - 1) // world)
        assert steps_per_pass == 1 and len(idx_lists) <= 1
        idx_lists = idx_lists + [None] * (steps_per_pass - len(idx_lists))
        args.batch = max(args.batch, (C5_BATCH + world - 1) // world)   # (sizes the pinned waveform buffers)
    elif args.workload == "testset":
        idx_lists = shard_indices(N_TEST, rank, world, args.batch)
        steps_per_pass = n_steps(N_TEST, world, args.batch)
        idx_lists = idx_lists + [None] * (steps_per_pass - len(idx_lists))   # ranks without a batch in the tail chunk still step
    else:
        # 4 different batches per rank: windows of `batch` consecutive test sentences starting 50 apart (wrapping)
        idx_lists = [[(rank * 25 + k * 50 + j) % N_TEST for j in range(args.batch)] for k in range(4)]
        steps_per_pass = 1

    def host_batch(idx):
        if idx is None:
            return None
        sent = [(c5_sents if c5_sents is not None else st["sentences"])[i] for i in idx]
        ib = synth.make_id_batch(sent, table)
        hb = {k: T(ib[k]).pin_memory() for k in ("word_tokens", "entry_ids", "pron_modified")}
        hb.update(B=len(sent), T_w=int(ib["word_tokens"].shape[1]), L_k=int(ib["L_k"]), P=int(ib["P"]), sent=sent)
        e = ib["entry_ids"]
        hb["live_gloss_rows"] = int(live_rows_of_entry[e[e >= 0]].sum())
        return hb

    batches = [host_batch(i) for i in idx_lists]
    dev_ids = None
    if args.input == "resident":
        dev_ids = [None if hb is None else {k: hb[k].to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")} for hb in batches]
    tens = None
    if args.input == "tensors":   # the reference's collated tensors, pinned on the host, uploaded inside every step
        tens = []
        for hb in batches:
            tb = synth.make_batch(hb["sent"], 1234)
            tens.append({k: T(v).pin_memory() for k, v in tb.items()})
    cap_frames = 1548                        # max_frames (egs/egs_bases/tts/base.yaml:45): capacity of the pinned waveform buffers
    pcm_host = [torch.empty(args.batch * cap_frames * hop, dtype=torch.int16).pin_memory() for _ in range(2)]   # flat: a step's [B, T_mel*hop] view is contiguous
    d2h_stream = torch.cuda.Stream(device=dev)   # the int16 waveforms leave on their own stream (copy engine) behind the vocoder stream

    gather_on = world > 1 and not args.no_gather
    gather_info = {"enabled": gather_on, "backend": ("gloo (one-device test hook)" if one_dev else "rccl") if world > 1 else None,
                   "calls": 0, "disabled_reason": None if gather_on or world == 1 else "--no-gather"}
    pipelined = not args.no_pipeline
    voc_stream = torch.cuda.Stream(device=dev, priority=args.voc_priority) if pipelined else None
    comm_stream = torch.cuda.Stream(device=dev) if gather_on else None
    ptr = lambda t: None if t is None else t.data_ptr()
    state = {"last": None, "k": 0}

    def run_step(k, mode, mdl=m, vc=voc):
        """one batch through the whole path; mode: table | resident | tensors (what crosses PCIe inside the step)"""
        hb = batches[k % len(batches)]
        cur = torch.cuda.current_stream()
        stream = cur.cuda_stream
        mel = lens = None
        T_mel = 0
        if hb is not None:
            B, T_w, L_k, P_ = hb["B"], hb["T_w"], hb["L_k"], hb["P"]
            if mode == "tensors":
                d = {kk: v.to(dev, non_blocking=True) for kk, v in tens[k % len(batches)].items()}   # 1.47 GB at B=60
                T_mel = mdl.ctx.text2mel_encode(ptr(d["word_tokens"]), ptr(d["keys"]), ptr(d["values"]), ptr(d["key_map"]),
                                                ptr(d["pinyin"]), ptr(d["pinyin_map"]), ptr(d["pron_modified"]), None, B, T_w,
                                                d["keys"].shape[2], d["pinyin"].shape[2], stream)
            else:
                if mode == "table":
                    d = {kk: hb[kk].to(dev, non_blocking=True) for kk in ("word_tokens", "entry_ids", "pron_modified")}
                else:
                    d = dev_ids[k % len(batches)]
                T_mel = mdl.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None,
                                                    B, T_w, L_k, P_, stream)
        shapes = None
        if gather_on:   # (B, T_mel) is known on the host as soon as encode has returned: the 2-int exchange starts HERE, on the communication
            # stream, and is read only after this batch's decode + vocoder have been enqueued (every rank enters, with or without a batch)
            with torch.cuda.stream(comm_stream):
                shapes = exchange_shapes(hb["B"] if hb is not None else 0, T_mel, dist, comm_device="cpu" if one_dev else None)
        if hb is not None:
            mel = torch.empty(B, T_mel, 80, device=dev)
            mdl.ctx.text2mel_decode(None, mel.data_ptr(), stream)     # z_p = NULL: the prior sample is drawn on the device
            lens = torch.empty(B, dtype=torch.int32, device=dev)
            mdl.ctx.fetch(abi.OUT_MEL_LENS, lens.data_ptr(), stream)

        def gather_step():   # the padded all-gather of this step's mels: behind the decoder on the communication stream; the host has
            # nothing left to enqueue for this batch when it reads the (long finished) shape exchange
            comm_stream.wait_stream(cur)
            with torch.cuda.stream(comm_stream):
                mel_all, lens_all, meta = gather_mels(mel, lens, dist, comm_device="cpu" if one_dev else None, shapes=shapes)
            gather_info["calls"] += 1
            gather_info["last_meta"] = meta.tolist()
            if mel_all is not None and mel is not None:
                mel.record_stream(comm_stream)
        if hb is None:
            if gather_on:
                gather_step()
            return None, None, 0
        out_stream = voc_stream if pipelined else cur
        if pipelined:
            # the acoustic model's launch-bound kernels leave most CUs idle: hand batch i's mel to the vocoder stream
            # and start text->mel of batch i+1 on this one (separate contexts, caller-owned mel/lens/wav buffers)
            voc_stream.wait_stream(cur)
        with torch.cuda.stream(out_stream):
            wav = vc.forward_batch(mel, lens)
            if mode != "resident":
                pcm = vc.to_int16(wav, lens)
                d2h_stream.wait_stream(out_stream)
                with torch.cuda.stream(d2h_stream):   # D2H of the int16 waveforms into pinned host memory
                    if T_mel <= cap_frames:
                        pcm_host[k & 1][:B * T_mel * hop].view(B, T_mel * hop).copy_(pcm, non_blocking=True)
                    else:
                        pcm.cpu()
                pcm.record_stream(d2h_stream)
        if pipelined:
            mel.record_stream(voc_stream)
            lens.record_stream(voc_stream)
        if gather_on:   # (the next step's text->mel does not wait for it: gathers are ordered among themselves on the communication stream)
            gather_step()
        state["last"] = (mel, lens, wav, hb, T_mel)
        return lens, wav, T_mel

    def timed_loop(n_steps_, n_warm, mode, sync_dist=True):
        # valid frames are counted AFTER the loop from the per-step length vectors (kept alive in a list): no bookkeeping kernels of the
        # harness run between the steps (round 2's `lens_acc += lens.sum()` queued two tiny torch kernels per step behind the
        # persistent vocoder workgroups: ~0.3 ms each on the text->mel stream)
        kept_lens = []
        for i in range(n_warm * steps_per_pass):  # identical to the timed loop body
            lens, _, _ = run_step(i, mode)
            if lens is not None:
                kept_lens.append(lens)
        torch.cuda.synchronize()
        state["warm_frames"] = int(sum(int(l.sum().item()) for l in kept_lens))
        kept_lens = []
        torch.cuda.synchronize()
        voc.ctx.timer_reset()
        if dist is not None and sync_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps_ * steps_per_pass):
            lens, _, _ = run_step(n_warm * steps_per_pass + i, mode)
            if lens is not None:
                kept_lens.append(lens)
        torch.cuda.synchronize()                 # includes the D2H copies into pinned memory
        if dist is not None and sync_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, int(sum(int(l.sum().item()) for l in kept_lens))

    # ---- fp16 range guard over the BENCHED batches: one guarded forward of each distinct batch before anything is timed.
    # The guarded kernel instantiations count every pre-activation the fp16 operands cannot represent (dtts_vocoder_clamped); the timed
    # loop then runs the release instantiations.  A non-zero count means DTTS_VOC_F16 is not valid for this workload.
    guard_info = None
    if args.precision == "f16":
        voc.ctx.vocoder_range_guard(True)
        clamped = g_frames = g_fw = 0
        for k, hb in enumerate(batches):
            lens_g, _, _ = run_step(k, args.input)   # (every rank steps through the same number of batches: the gather stays matched)
            if lens_g is None:
                continue
            torch.cuda.synchronize()
            clamped += voc.ctx.vocoder_clamped(torch.cuda.current_stream().cuda_stream)
            g_frames += int(lens_g.sum().item())
            g_fw += 1
        voc.ctx.vocoder_range_guard(False)
        guard_info = {"clamped_activations": int(clamped), "guarded_forwards": g_fw, "mel_frames": g_frames,
                      "what": "dtts_vocoder_range_guard on for one forward of every distinct benched batch, before the timed loop"}
    elapsed, frames_rank = timed_loop(args.steps, args.warmup, args.input)
    main_warm_frames = state["warm_frames"]
    conv_ms, conv_launches = voc.ctx.timer_read(abi.TIMER_VOC_CONV)
    assert state["last"] is None or (torch.isfinite(state["last"][2]).all() and float(state["last"][2].abs().max()) <= 1.0)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_dev else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        f = torch.tensor([frames_rank], dtype=torch.int64, device="cpu" if one_dev else dev)
        dist.all_reduce(f)
        frames_total = int(f.item())
    else:
        frames_total = frames_rank

    # ---- after the timed region, rank 0 only, nothing else on the GPU
    side = stages = iso = modes = voc2 = other = None
    if rank == 0 and not args.no_side and state["last"] is not None:
        gather_save, gather_on = gather_on, False       # the side measurements are single-rank
        side = {}
        for name, mode, n in (("inputs_resident", "resident", 6), ("tensor_api_incl_h2d", "tensors", 3)):
            if mode == args.input:
                continue
            if mode == "resident" and dev_ids is None:
                dev_ids = [None if hb is None else {k: hb[k].to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")} for hb in batches]
            if mode == "tensors" and tens is None:
                tens = []
                for hb in batches[:2]:
                    tens.append({k: T(v).pin_memory() for k, v in synth.make_batch(hb["sent"], 1234).items()})
                keep_batches, batches = batches, batches[:2]
            dt, fr = timed_loop(n, 1, mode, sync_dist=False)
            what = "ids already in HBM, fp32 waveform left in HBM (no PCIe traffic in the step)"
            if mode == "tensors":
                h2d = sum(v.numel() * v.element_size() for v in tens[0].values())
                what = (f"reference API: keys/values/key_map/pinyin tensors uploaded per batch from pinned host memory "
                        f"({h2d / 1e9:.2f} GB H2D per step) + int16 waveform D2H")
                batches = keep_batches
                tens = None
            side[name] = {"mel_frames_per_s": fr / dt, "ms_per_step": dt / (n * steps_per_pass) * 1e3, "steps": n, "what": what}
        # per-stage rates and the rooflines SURVEY.md 8(d) names for the other two stages, one stream, 3 repetitions
        m.ctx.timer_enable(abi.TIMER_S2PA)
        for tm in (abi.TIMER_STAGE_ENCODER, abi.TIMER_STAGE_DICT_ENCODER, abi.TIMER_STAGE_FVAE):
            m.ctx.timer_enable(tm)
        voc.ctx.timer_enable(abi.TIMER_STAGE_HIFIGAN)
        m.ctx.timer_reset()
        voc.ctx.timer_reset()
        hb = next(b for b in batches if b is not None)
        d = {k: hb[k].to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")}
        stream = torch.cuda.current_stream().cuda_stream
        # one untimed pass (it also sizes the caller-owned mel / length buffers, which the timed passes reuse: the same batch every time, so
        # no allocator call sits between the events), then the MEDIAN of five — round 5 averaged three passes including the first one, and an
        # allocator round trip inside the decode bracket showed up as +0.4 ms on one box
        reps_e, reps_d, reps_v = [], [], []
        mel_i = lens_i = None
        for rep in range(6):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            T_m = m.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None,
                                            hb["B"], hb["T_w"], hb["L_k"], hb["P"], stream)
            ev[1].record()
            if mel_i is None or mel_i.shape[1] != T_m:
                mel_i = torch.empty(hb["B"], T_m, 80, device=dev)
                lens_i = torch.empty(hb["B"], dtype=torch.int32, device=dev)
            m.ctx.text2mel_decode(None, mel_i.data_ptr(), stream)
            m.ctx.fetch(abi.OUT_MEL_LENS, lens_i.data_ptr(), stream)
            ev[2].record()
            voc.forward_batch(mel_i, lens_i)
            ev[3].record()
            torch.cuda.synchronize()
            if rep == 0:
                m.ctx.timer_reset()
                voc.ctx.timer_reset()
                continue
            reps_e.append(ev[0].elapsed_time(ev[1]))
            reps_d.append(ev[1].elapsed_time(ev[2]))
            reps_v.append(ev[2].elapsed_time(ev[3]))
        med = lambda v: sorted(v)[len(v) // 2]
        ms_enc, ms_dec, ms_voc = med(reps_e), med(reps_d), med(reps_v)
        s2pa_ms, s2pa_n = m.ctx.timer_read(abi.TIMER_S2PA)
        per_call = lambda ctx, which: (lambda ms, n: ms / max(n, 1))(*ctx.timer_read(which))
        ref_names = {"encoder": per_call(m.ctx, abi.TIMER_STAGE_ENCODER), "dict_encoder": per_call(m.ctx, abi.TIMER_STAGE_DICT_ENCODER),
                     "fvae": per_call(m.ctx, abi.TIMER_STAGE_FVAE), "hifigan": per_call(voc.ctx, abi.TIMER_STAGE_HIFIGAN)}
        fr = int(lens_i.sum().item())
        dec_tf = FLOP_PER_FRAME_DECODER * hb["B"] * T_m / (ms_dec * 1e-3) / 1e12
        stages = {"isolated": True, "batch": "first batch of this rank", "mel_frames_per_batch": fr,
                  "reference_profile_infer_timers_ms": ref_names,   # utils.Timer names of modules/dict_tts/model.py:50,57,86, vocoders/hifigan.py:59
                  "how": "one untimed pass, then the median of five passes of encode | decode | vocoder between torch events on one stream",
                  "text2mel": {"ms": ms_enc + ms_dec, "encode_ms": ms_enc, "decode_ms": ms_dec,
                               "mel_frames_per_s": fr / ((ms_enc + ms_dec) * 1e-3)},
                  "vocoder": {"ms": ms_voc, "mel_frames_per_s": fr / (ms_voc * 1e-3)},
                  "end_to_end_serial": {"ms": ms_enc + ms_dec + ms_voc, "mel_frames_per_s": fr / ((ms_enc + ms_dec + ms_voc) * 1e-3)},
                  "mel_decoder_roofline": {"bound": "mfma", "achieved": dec_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                           "frac": dec_tf / PEAK_BF16_TFLOPS,
                                           "note": "A8-A10, 4,691,968 FLOP per padded mel frame, against the bf16 MFMA peak BASELINE.md §4 names"}}
        if s2pa_ms > 0:
            # bytes the kernel MOVES on this path: the resident table holds the projected rows K = key Wk^T, V = value Wv^T
            # (hidden_size = 192 floats each): 2 x 768 B per live gloss row (include/dicttts_hip.h: dtts_dict_table_upload)
            row_bytes = 2 * 4 * 192
            gbs = row_bytes * hb["live_gloss_rows"] / (s2pa_ms / max(s2pa_n, 1) * 1e-3) / 1e9
            stages["s2pa_roofline"] = {"bound": f"hbm (nominal) - the kernel is LATENCY-bound: {s2pa_ms / max(s2pa_n, 1) * 1e3:.1f} us per launch, ~15 live rows per word, "
                                                "one dependent chain entry -> offsets -> key_map -> rows -> reductions per workgroup (DESIGN.md 3.3)",
                                       "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                       "kernel": "dtts::s2pa_kernel<1, 3, true> (resident table of PRE-PROJECTED rows, gathered by entry id; 16 lanes per row)",
                                       "avg_launch_ms": s2pa_ms / max(s2pa_n, 1), "avg_launch_us": s2pa_ms / max(s2pa_n, 1) * 1e3,
                                       "algorithmic_bytes": f"{row_bytes} B x live gloss rows of the batch's entries (fp32 K + V, 192 wide each; "
                                                            "the tensor API reads 6144 B per row: raw 768-wide key + value)",
                                       "bytes_per_launch": row_bytes * hb["live_gloss_rows"],
                                       "live_gloss_rows": hb["live_gloss_rows"]}
        # BASELINE.md §4 row "S2PA (A3), tensor API": the drop-in API's own kernel (s2pa_kernel<3, .>: raw 768-wide key + value rows of the
        # collated tensors, re-associated logits key . (Wk^T q)) on tensors ALREADY RESIDENT in HBM (the PCIe-inclusive figure is side.tensor_api_incl_h2d)
        tb0 = {k: T(v).to(dev) for k, v in synth.make_batch(hb["sent"], 1234).items()}
        live_t = int((tb0["key_map"] != 0).sum().item())
        m.ctx.timer_reset()
        for _ in range(4):
            m.ctx.text2mel_encode(ptr(tb0["word_tokens"]), ptr(tb0["keys"]), ptr(tb0["values"]), ptr(tb0["key_map"]), ptr(tb0["pinyin"]),
                                  ptr(tb0["pinyin_map"]), ptr(tb0["pron_modified"]), None, hb["B"], hb["T_w"], tb0["keys"].shape[2],
                                  tb0["pinyin"].shape[2], stream)
        torch.cuda.synchronize()
        s2t_ms, s2t_n = m.ctx.timer_read(abi.TIMER_S2PA)
        if s2t_ms > 0:
            per = s2t_ms / max(s2t_n, 1)
            gbs_t = 6144.0 * live_t / (per * 1e-3) / 1e9
            stages["s2pa_tensor_roofline"] = {
                "bound": "hbm (nominal)" + ("" if gbs_t / PEAK_HBM_GBS >= 0.5 else f" - LATENCY-bound: {per * 1e3:.1f} us per launch, one dependent chain key_map -> live-row list -> "
                                                                                   "rows -> reductions per word and ~15 live rows of 6 KB per word (DESIGN.md 3.3)"),
                "achieved": gbs_t, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs_t / PEAK_HBM_GBS,
                "kernel": "dtts::s2pa_kernel<3, .> (tensor API: collated keys / values [B,T_w,L_k,768] resident in HBM)",
                "avg_launch_ms": per, "avg_launch_us": per * 1e3, "launches": s2t_n,
                "algorithmic_bytes": "6,144 B (fp32 key row + value row, 768 wide each) x live gloss rows (key_map != 0) of the batch (BASELINE.md §4)",
                "bytes_per_launch": 6144 * live_t, "live_gloss_rows": live_t,
                "padded_bytes_per_launch": int(6144 * hb["B"] * hb["T_w"] * tb0["keys"].shape[2]),
                "note": "the padded tensors hold 6,144 x B x T_w x L_k bytes; the kernel reads only the live rows"}
        del tb0
        m.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None, hb["B"], hb["T_w"], hb["L_k"], hb["P"], stream)   # (back on the table path)
        # A2 (BASELINE.md §4 row "Text encoder blocks, fp32 MFMA, 157.3 TF"): SURVEY 8d counts 8 * (2,064,384 + 768 T_w) FLOP per word
        # token for the two 4-layer encoders; time = the 'dict_encoder' span (embedding + both encoders + S2PA: slightly conservative)
        a2_flop = 8.0 * (2_064_384 + 768 * hb["T_w"]) * hb["B"] * hb["T_w"]
        if ref_names["dict_encoder"] > 0:
            a2_tf = a2_flop / (ref_names["dict_encoder"] * 1e-3) / 1e12
            stages["encoder_blocks_roofline"] = {"bound": "mfma (fp32-grade arithmetic keeps the integer durations: bf16 three-piece operands, 6 MFMA products; peak = the fp32 matrix rate BASELINE.md names, the 6-product form's own ceiling is 2500 / 6 = 417)", "achieved": a2_tf, "peak": 157.3,
                                                 "unit": "TFLOP/s", "frac": a2_tf / 157.3, "span_ms": ref_names["dict_encoder"],
                                                 "flop": a2_flop, "padded_word_tokens": hb["B"] * hb["T_w"],
                                                 "note": "A2 rows of SURVEY 8a over the padded batch; ~90 launches of 5-25 us bound by per-kernel round trips "
                                                         "(staging 3-4 us, epilogue 1-1.5 us) and one wave's MFMA chain (60 utterances x 32-row tiles; "
                                                         "tools/c1d_phase_prof.py), not by the matrix rate"}
        # how often the round() discontinuity of add_dur (model.py:78) is in play: words of the GPU's own dur within 5e-5 of a .5 tie
        dur_t = torch.empty(hb["B"], hb["T_w"], device=dev)
        m.ctx.fetch(abi.OUT_DUR, dur_t.data_ptr(), stream)
        torch.cuda.synchronize()
        dv = torch.exp(dur_t.double()) - 1.0
        wmask = (d["word_tokens"] > 0)
        fracd = (dv - torch.floor(dv) - 0.5).abs()
        stages["duration_ties"] = {"words": int(wmask.sum().item()), "within_5e-5_of_half": int(((fracd < 5e-5) & wmask).sum().item()),
                                   "within_1e-3_of_half": int(((fracd < 1e-3) & wmask).sum().item()),
                                   "note": "a word whose exp(dur)-1 sits this close to x.5 may round either way between two fp32 summation "
                                           "orders (GPU vs CPU thread counts); tests/test_gpu_parity.py::test_config2 prints n_ties / n_flips"}
        # ---- the two single-utterance shapes of BASELINE.json: configs[0] (one Biaobei sentence, B = 1) and configs[3] (long
        # form: 1,000 characters, teacher-forced 5 frames per character -> ~5 k mel frames, B = 1), serial text->mel -> vocoder on one stream
        # with the resident table; 5 repetitions after 2 warm-ups.  Small grids: a launch lasts as long as its slowest workgroup's tile chain.
        def single_utterance(sent, frames_per_char=None):
            ib1 = synth.make_id_batch([sent], table)
            d1 = {k: T(ib1[k]).to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")}
            m2w = None
            if frames_per_char:
                m2w_t = T(synth.teacher_mel2word(ib1["word_tokens"], frames_per_char, frames_per_char)).to(dev)
                m2w = (m2w_t.data_ptr(), int(m2w_t.shape[1]))
            t2m = vo = 0.0
            reps = 5
            for it in range(reps + 2):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                T_m = m.ctx.text2mel_encode_ids(ptr(d1["word_tokens"]), ptr(d1["entry_ids"]), ptr(d1["pron_modified"]), m2w, 1,
                                                int(ib1["word_tokens"].shape[1]), ib1["L_k"], ib1["P"], stream)
                mel1 = torch.empty(1, T_m, 80, device=dev)
                m.ctx.text2mel_decode(None, mel1.data_ptr(), stream)
                lens1 = torch.empty(1, dtype=torch.int32, device=dev)
                m.ctx.fetch(abi.OUT_MEL_LENS, lens1.data_ptr(), stream)
                ev[1].record()
                voc.forward_batch(mel1, lens1)
                ev[2].record()
                torch.cuda.synchronize()
                if it >= 2:
                    t2m += ev[0].elapsed_time(ev[1]) / reps
                    vo += ev[1].elapsed_time(ev[2]) / reps
            fr1 = int(lens1.sum().item())
            return {"characters": len(sent), "mel_frames": fr1, "audio_s": fr1 * hop / 22050.0, "text2mel_ms": t2m, "vocoder_ms": vo, "ms": t2m + vo,
                    "mel_frames_per_s": fr1 / ((t2m + vo) * 1e-3), "rtf": (t2m + vo) * 1e-3 / (fr1 * hop / 22050.0)}
        stages["b1"] = {**single_utterance(st["sentences"][0]), "what": "BASELINE configs[0] on the GPU: one Biaobei sentence, B = 1, predicted durations"}
        stages["long_form"] = {**single_utterance([w for sn in st["sentences"] for w in sn][:1000], 5),
                               "what": "BASELINE configs[3]: 1,000 characters (T_w = 1,002), teacher-forced 5 frames per character, B = 1"}
        m.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None, hb["B"], hb["T_w"], hb["L_k"], hb["P"], stream)   # (back on the benched batch)
        # the same vocoder kernels on the last batch with nothing else on the GPU (in the timed region they share the CUs
        # with the next batch's text->mel kernels), and the all-bf16 mode beside the default one
        mel_l, lens_l = state["last"][0], state["last"][1]
        fr_l = int(lens_l.sum().item())

        def isolated(v):
            v.ctx.timer_enable(abi.TIMER_VOC_CONV)
            v.forward_batch(mel_l, lens_l)
            torch.cuda.synchronize()
            v.ctx.timer_reset()
            for _ in range(3):
                v.forward_batch(mel_l, lens_l)
            torch.cuda.synchronize()
            ms, n = v.ctx.timer_read(abi.TIMER_VOC_CONV)
            return ms / 3, n // 3
        iso_ms, iso_n = isolated(voc)
        iso = (iso_ms, iso_n, fr_l)
        # meets_waveform_gate is MEASURED below (cpu_baseline leg: oracle waveforms of utterances 0..2) — null when that leg is off
        modes = {args.precision: {"vocoder_kernel_ms_per_forward": iso_ms, "vocoder_mel_frames_per_s": fr_l / (iso_ms * 1e-3),
                                  "meets_waveform_gate": None, "range_guard": guard_info}}
        if args.precision == "f16":   # the fp16 decision (include/dicttts_hip.h): static bound + the always-on conv_post detector over EVERY forward of this run
            torch.cuda.synchronize()
            modes["f16"]["fp16_validity"] = {"static": voc.fp16_status, "worst_case_bound_mel6": voc.fp16_bound[0], "rms_estimate_mel6": voc.fp16_bound[1],
                                             "nonfinite_outputs_all_forwards": int(voc.ctx.vocoder_nonfinite())}
        other = "bf16" if args.precision != "bf16" else "f16"
        voc2 = vocoder.HifiGAN(state_dict=voc_sd, config=synth.hifigan_config(), precision=abi.VOC_PRECISIONS[other])
        o_ms, _ = isolated(voc2)
        modes[other] = {"vocoder_kernel_ms_per_forward": o_ms, "vocoder_mel_frames_per_s": fr_l / (o_ms * 1e-3),
                        "meets_waveform_gate": None}
        gather_on = gather_save

    if rank == 0:
        value = frames_total / elapsed
        samples = value * hop
        n_timed = args.steps * steps_per_pass
        achieved = FLOP_PER_FRAME_VOCODER * frames_rank / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # HBM bytes of the same kernel family: PMC counters cannot be read from inside the process, so the committed
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE result (profiles/r*_pmc_traffic.json of the newest round, corrected
        # as MI355X_MICROARCH.md prescribes) is scaled by this run's frame count; null if the file is absent
        traffic = traffic_src = None
        import glob
        tps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        if tps and conv_launches:
            with open(tps[-1]) as f:
                tj = json.load(f)
            if tj.get("vocoder_precision", "bf16") == args.precision:
                traffic = tj["hbm_bytes_per_mel_frame"] * frames_rank / conv_launches
                traffic_src = f"profiles/{os.path.basename(tps[-1])} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; bytes per launch, avg)"
        last = state["last"]
        shapes = sorted({(b["B"], b["T_w"], b["L_k"]) for b in batches if b is not None})
        out = {
            "metric": "mel-frames/sec, end-to-end text->mel->wav (audio-samples/sec = 256x; RTF reported alongside)",
            "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.workload == "rotating" else "strong", "vs_baseline": None,
            "dtype": {"f16": "f16 (vocoder ResBlocks fp16 MFMA, serial convolutions bf16 hi/lo split; acoustic model fp32-grade: fp32 MFMA / bf16 three-piece operands x 6 products)",
                      "bf16": "bf16", "bf16x3": "bf16x3"}[args.precision],
            "data": "synthetic (random-init weights of the real architecture, Biaobei sentence/dictionary structure)",
            "config": {"workload": ("BASELINE configs[1]: Biaobei batch=60 per GPU, full Dict-TTS encoder + FVAE decoder + HifiGAN, predicted "
                                    "durations (~22 frames/char); the step rotates over 4 different batches of the 200-sentence test set"
                                    if args.workload == "rotating" else
                                    "BASELINE configs[2]: one step = all 200 test sentences, utterance i -> rank i mod N in batches <= 60 "
                                    "(tts_base.py:148-151)" if args.workload == "testset" else
                                    "BASELINE configs[4]: dictionary stress — all 7,030 zh-dict.json entries resident, one step = ONE batch of 128 "
                                    "mixed-length utterances (6..60 characters, heteronyms x5), utterance i -> rank i mod N (tts_base.py:148-151)"),
                       "utterances_per_gpu_per_batch": args.batch, "batches_per_step": steps_per_pass,
                       "distinct_batches": len([b for b in batches if b is not None]), "batch_shapes_B_Tw_Lk": shapes,
                       "mel_frames_per_step_per_gpu": frames_rank // max(args.steps, 1),
                       "all_forwards": {"n": (args.warmup + args.steps) * steps_per_pass + (guard_info["guarded_forwards"] if guard_info else 0),   # what a profiler attached to this run sees
                                        "mel_frames": main_warm_frames + frames_rank + (guard_info["mel_frames"] if guard_info else 0)},   # (incl. the range-guard pass)
                       "step_includes": {"table": "H2D of the batch's ids + device prior sample + int16 conversion + D2H of the int16 waveforms "
                                                  "(dictionary resident in HBM: dtts_dict_table_upload, once)",
                                         "resident": "ids resident in HBM; fp32 waveform left in HBM",
                                         "tensors": "reference API: keys/values tensors uploaded per step + int16 waveform D2H"}[args.input],
                       "parallelism": f"dp{world}" + ("+allgather(mel)" if gather_on else ""),
                       "streams": "2 (vocoder of batch i overlaps text->mel of batch i+1)" if pipelined else "1",
                       "acoustic_dtype": "f32-grade (fp32 MFMA; word-encoder convolutions as 3 bf16 pieces x 6 products)", "vocoder_dtype": args.precision},
            "audio_samples_per_sec": samples, "rtf": 22050.0 / samples,
            "mel_allgather": gather_info,
            "ranks_seen": seen, "distinct_devices": (len({(r["device_index"], r["device_uuid"]) for r in seen}) if seen else 1),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "data_ceiling": {"peak": DATA_CEILING_F16_TFLOPS, "frac": achieved / DATA_CEILING_F16_TFLOPS,
                                          "what": "the matrix rate a bare MFMA loop sustains with data-carrying fp16 operands (the chip throttles MFMA issue with "
                                                  "operand toggling: 1.67 PF against 2.34-2.48 PF with zeros; tools/micro/mfma_peak, profiles/*_mfma_ceiling.txt) — "
                                                  "context for `frac`, which stays priced against the 2.5 PF dense peak"},
                         "kernel": "dtts::vconv_kernel<*> + dtts::vpair_kernel<256|128,*> + dtts::rblock_kernel<*> (every HifiGAN convolution; "
                                   f"{conv_launches // max(n_timed, 1)} launches/forward)",
                         "launches": conv_launches, "avg_launch_ms": conv_ms / max(conv_launches, 1),
                         "kernel_ms_per_step": conv_ms / max(args.steps, 1),
                         "algorithmic_flop_per_mel_frame": FLOP_PER_FRAME_VOCODER,
                         "note": "achieved = algorithmic FLOPs (one product per MAC; the split-operand serial convolutions issue three) / "
                                 "hipEvent time of these kernels inside the timed region, where they share the CUs with the next batch's "
                                 "text->mel kernels; 'isolated' = the same kernels on the last batch alone"},
        }
        if iso is not None and iso[0] > 0:
            ia = FLOP_PER_FRAME_VOCODER * iso[2] / (iso[0] * 1e-3) / 1e12
            out["roofline"]["isolated"] = {"achieved": ia, "frac": ia / PEAK_BF16_TFLOPS, "frac_of_data_ceiling": ia / DATA_CEILING_F16_TFLOPS,
                                           "kernel_ms_per_forward": iso[0], "launches": iso[1], "mel_frames": iso[2],
                                           "what": "three forwards of the LAST timed batch alone on the GPU; achieved / frac are priced on THAT batch's own valid "
                                                   "frame count (mel_frames), not on the run's average batch"}
        if side:
            out["side"] = side
        if modes:
            out["vocoder_modes"] = modes
        if stages is not None:
            out["stages"] = stages
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(torch, np, synth, voc, args.cpu_baseline, args.cpu_budget)
            kept_all = cb.pop("_kept")
            out["cpu_baseline"] = cb
            if modes:   # the gate of each listed mode, measured against the oracle's waveforms (and, for f16, a clean range guard)
                wc = cb["waveform_check"]
                modes[args.precision]["waveform_check"] = {k: wc[k] for k in ("rms_diff", "abs_rms_delta", "frames", "pass")}
                modes[args.precision]["meets_waveform_gate"] = bool(wc["pass"] and (guard_info is None or guard_info["clamped_activations"] == 0) and
                                                                    modes[args.precision].get("fp16_validity", {}).get("nonfinite_outputs_all_forwards", 0) == 0)
                wc2 = waveform_check(np, voc2, kept_all)
                modes[other]["waveform_check"] = {k: wc2[k] for k in ("rms_diff", "abs_rms_delta", "frames", "pass")}
                modes[other]["meets_waveform_gate"] = bool(wc2["pass"])
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
