#!/usr/bin/env python3
"""Headline benchmark of the Dict-TTS inference hot path on MI355X (BASELINE.json: mel-frames/sec + audio-samples/sec
(RTF) per GPU, Biaobei batch=60).

One "step" = one pass of the whole hot path over one batch of 60 synthetic Biaobei utterances per GPU:
    S2PA dictionary encoder -> duration predictor -> length regulator (incl. the one T_mel host sync)
    -> prior flow + FVAE decoder -> HifiGAN (bf16 MFMA)            text ids + gloss embeddings in HBM -> waveform in HBM
Inputs are resident in HBM before the timed region (the PCIe-inclusive figure is discussed in DESIGN.md).
Data: synthetic (counter-based weights / gloss embeddings, real Biaobei sentence + dictionary structure); durations
are PREDICTED by the duration predictor (its output bias is set so that the mean is ~22 frames per character).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Weak scaling: every rank processes its own batch of 60 (utterances r*60.. of the
200-sentence test set, wrapping around); with N > 1 the mels are all-gathered over RCCL (overlapped with the vocoder).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FRAME_VOCODER = 614_105_088   # 2*MAC of HifiGanGenerator per mel frame
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
DUR_BIAS = 3.09                        # exp(softplus(3.09)) - 1 ~= 22 frames per word


def cpu_baseline(torch, np, synth, n_utt=20, max_seconds=25.0):
    """the CPU oracle (our restatement of the reference, oracle/*.py) timed on the host cores: reference-faithful
    protocol, B=1 per utterance, model forward + one spec2wav per utterance (tasks/tts/dict_tts.py:179-255)"""
    from oracle import dict_tts_ref as ref
    from oracle import hifigan_ref as href
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    sd_np = synth.dict_tts_state_dict(1234)
    sd_np["dur_predictor.linear.0.bias"] = np.array([DUR_BIAS], np.float32)
    sd = href.fold_weight_norm({k: T(v) for k, v in sd_np.items()})
    hsd = href.fold_weight_norm({k: T(v) for k, v in synth.hifigan_state_dict(1234).items()})
    cfg = synth.hifigan_config()
    st = synth.biaobei_struct()
    frames, t_total, done = 0, 0.0, 0
    for i in range(n_utt + 1):  # first one is the warm-up
        b = {k: T(v) for k, v in synth.make_batch([st["sentences"][i % 200]], 1234).items()}
        t0 = time.perf_counter()
        r = ref.forward_infer(sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                              b["pron_modified"], z_p=lambda B, T4: torch.randn(B, 16, T4))
        wav = href.spec2wav(hsd, cfg, r["mel_out"][0].numpy())
        dt = time.perf_counter() - t0
        assert wav.numel() == r["mel_out"].shape[1] * 256
        if i == 0:
            continue
        frames += int(r["mel_out"].shape[1])
        t_total += dt
        done += 1
        if t_total > max_seconds:
            break
    return {"value": frames / t_total, "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "sample": f"{done} Biaobei utterances at B=1 (text->mel->wav, {frames} frames, {t_total:.1f} s), torch CPU fp32 oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=60)
    ap.add_argument("--precision", choices=["f16", "bf16", "bf16x3"], default="f16",
                    help="vocoder arithmetic (include/dicttts_hip.h): f16 = the waveform-exact default")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of mels when --gpus > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dict-table", action="store_true",
                    help="resident dictionary table: batches carry entry ids instead of keys/values tensors (SURVEY 8f-1)")
    ap.add_argument("--phases", action="store_true", help="debug: per-phase device/host times on stderr")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the vocoder on the text->mel stream (default: vocoder of batch i on a second HIP stream, "
                         "overlapping text->mel of batch i+1; every batch is still fully processed inside the timed region)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from dict_tts_amd import abi, model, synth, vocoder
    from dict_tts_amd.shard import shard_indices

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # test hook (1-GPU box): DTTS_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 and uses gloo, so that the N>1 control
    # flow (rendezvous, barriers, max-over-ranks timing, frame all-reduce) can be exercised without N GPUs
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

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    # ---- weights (random-init of the real architecture) and the acoustic model / vocoder behind the reference APIs
    sd_np = synth.dict_tts_state_dict(1234)
    sd_np["dur_predictor.linear.0.bias"] = np.array([DUR_BIAS], np.float32)
    m = model.PortaSpeech_dict(hparams={})
    m.load_state_dict({k: T(v) for k, v in sd_np.items()})
    prec = abi.VOC_PRECISIONS[args.precision]
    voc = vocoder.HifiGAN(state_dict={k: T(v) for k, v in synth.hifigan_state_dict(1234).items()},
                          config=synth.hifigan_config(), precision=prec)
    voc.ctx.timer_enable(abi.TIMER_VOC_CONV)
    # ---- this rank's batch, resident in HBM
    st = synth.biaobei_struct()
    sent = [st["sentences"][(rank * args.batch + i) % len(st["sentences"])] for i in range(args.batch)]
    if args.dict_table:
        table = synth.dict_table(1234)
        m.upload_dict_table(table)
        ib = synth.make_id_batch(sent, table)
        batch = {k: T(v).to(dev) for k, v in ib.items() if k not in ("L_k", "P")}
        B, T_w = batch["word_tokens"].shape
        L_k, P_ = ib["L_k"], ib["P"]
    else:
        batch = {k: T(v).to(dev) for k, v in synth.make_batch(sent, 1234).items()}
        B, T_w = batch["word_tokens"].shape
        L_k, P_ = batch["keys"].shape[2], batch["pinyin"].shape[2]
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    z_all = torch.randn(B, 16, 4096, generator=gen).to(dev)  # prior noise, sliced to T_mel/4 each step
    CAP = 1548                                                # max_frames (egs/egs_bases/tts/base.yaml:45)
    gather_on = world > 1 and not args.no_gather and not one_dev
    gather_state = [gather_on, None]   # [enabled, reason it was switched off]
    if gather_on:
        mel_pad = torch.zeros(B, CAP, 80, device=dev)
        mel_all = torch.empty(world * B, CAP, 80, device=dev)
        comm_stream = torch.cuda.Stream(device=dev)

    pipelined = not args.no_pipeline and not args.phases
    voc_prio = int(os.environ.get("DTTS_BENCH_VOC_PRIO", "-1"))
    voc_stream = torch.cuda.Stream(device=dev, priority=voc_prio) if pipelined else None

    phase_log = []

    def run_step():
        # encode (one host sync: T_mel) -> decode -> vocoder; z_p sliced from the resident noise
        stream = torch.cuda.current_stream().cuda_stream
        if args.phases:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            h0 = time.perf_counter()
            ev[0].record()
        ptr = lambda t: t.data_ptr()
        if args.dict_table:
            T_mel = m.ctx.text2mel_encode_ids(ptr(batch["word_tokens"]), ptr(batch["entry_ids"]), ptr(batch["pron_modified"]),
                                              None, B, T_w, L_k, P_, stream)
        else:
            T_mel = m.ctx.text2mel_encode(ptr(batch["word_tokens"]), ptr(batch["keys"]), ptr(batch["values"]),
                                          ptr(batch["key_map"]), ptr(batch["pinyin"]), ptr(batch["pinyin_map"]),
                                          ptr(batch["pron_modified"]), None, B, T_w, L_k, P_, stream)
        if args.phases:
            h1 = time.perf_counter()
            ev[1].record()
        z = z_all[:, :, : T_mel // 4].contiguous()
        mel = torch.empty(B, T_mel, 80, device=dev)
        m.ctx.text2mel_decode(z.data_ptr(), mel.data_ptr(), stream)
        if args.phases:
            h2 = time.perf_counter()
            ev[2].record()
        lens = torch.empty(B, dtype=torch.int32, device=dev)
        m.ctx.fetch(abi.OUT_MEL_LENS, lens.data_ptr(), stream)
        work = None
        if gather_state[0]:
            n_cp = min(T_mel, CAP)
            mel_pad[:, :n_cp] = mel[:, :n_cp]
            mel_pad[:, n_cp:].zero_()
            comm_stream.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(comm_stream):
                    work = dist.all_gather_into_tensor(mel_all, mel_pad, async_op=True)
            except Exception as e:   # the optional collective must not take the measurement down with it
                gather_state[0] = False
                gather_state[1] = f"{type(e).__name__}: {e}"[:160]
                print(f"[bench] rank {rank}: mel all-gather disabled: {gather_state[1]}", file=sys.stderr)
        if pipelined:
            # the acoustic model's launch-bound kernels leave most CUs idle: hand batch i's mel to the vocoder stream
            # and start text->mel of batch i+1 on this one (separate contexts, caller-owned mel/lens/wav buffers)
            voc_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(voc_stream):
                wav = voc.forward_batch(mel, lens)
            mel.record_stream(voc_stream)
            lens.record_stream(voc_stream)
        else:
            wav = voc.forward_batch(mel, lens)
        if work is not None:
            work.wait()
        if args.phases:
            h3 = time.perf_counter()
            ev[3].record()
            phase_log.append((ev, (h0, h1, h2, h3)))
        nonlocal_mel[0] = mel
        return lens, wav, T_mel

    nonlocal_mel = [None]
    lens_acc = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(args.warmup):  # identical to the timed loop body (torch lazily loads its reduce/add kernels on first use)
        lens, wav, T_mel = run_step()
        lens_acc += lens.sum()
    lens_acc.zero_()
    torch.cuda.synchronize()
    voc.ctx.timer_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames_rank = 0
    for _ in range(args.steps):
        lens, wav, T_mel = run_step()
        lens_acc += lens.sum()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    frames_rank = int(lens_acc.item())
    conv_ms, conv_launches = voc.ctx.timer_read(abi.TIMER_VOC_CONV)
    stages = None
    if rank == 0:
        # outside the timed region, nothing else on the GPU: per-stage rates and the rooflines SURVEY.md 8(d) names
        # for the other two stages (S2PA: HBM-bound; mel decoder: fp32 MFMA), one stream, 3 repetitions
        m.ctx.timer_enable(abi.TIMER_S2PA)
        m.ctx.timer_reset()
        ms_enc = ms_dec = ms_voc = 0.0
        stream = torch.cuda.current_stream().cuda_stream
        ptr = lambda t: t.data_ptr()
        for _ in range(3):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            if args.dict_table:
                T_m = m.ctx.text2mel_encode_ids(ptr(batch["word_tokens"]), ptr(batch["entry_ids"]), ptr(batch["pron_modified"]),
                                                None, B, T_w, L_k, P_, stream)
            else:
                T_m = m.ctx.text2mel_encode(ptr(batch["word_tokens"]), ptr(batch["keys"]), ptr(batch["values"]),
                                            ptr(batch["key_map"]), ptr(batch["pinyin"]), ptr(batch["pinyin_map"]),
                                            ptr(batch["pron_modified"]), None, B, T_w, L_k, P_, stream)
            ev[1].record()
            z_i = z_all[:, :, : T_m // 4].contiguous()
            mel_i = torch.empty(B, T_m, 80, device=dev)
            m.ctx.text2mel_decode(z_i.data_ptr(), mel_i.data_ptr(), stream)
            lens_i = torch.empty(B, dtype=torch.int32, device=dev)
            m.ctx.fetch(abi.OUT_MEL_LENS, lens_i.data_ptr(), stream)
            ev[2].record()
            voc.forward_batch(mel_i, lens_i)
            ev[3].record()
            torch.cuda.synchronize()
            ms_enc += ev[0].elapsed_time(ev[1]) / 3
            ms_dec += ev[1].elapsed_time(ev[2]) / 3
            ms_voc += ev[2].elapsed_time(ev[3]) / 3
        s2pa_ms, s2pa_n = m.ctx.timer_read(abi.TIMER_S2PA)
        fr = int(lens_i.sum().item())
        if args.dict_table:
            gloss_rows = None
        else:
            gloss_rows = int((batch["key_map"] != 0).sum().item())   # rows the softmax does not mask = rows that must be read
        stages = {"isolated": True, "mel_frames_per_batch": fr,
                  "text2mel": {"ms": ms_enc + ms_dec, "encode_ms": ms_enc, "decode_ms": ms_dec,
                               "mel_frames_per_s": fr / ((ms_enc + ms_dec) * 1e-3)},
                  "vocoder": {"ms": ms_voc, "mel_frames_per_s": fr / (ms_voc * 1e-3)},
                  "end_to_end_serial": {"ms": ms_enc + ms_dec + ms_voc, "mel_frames_per_s": fr / ((ms_enc + ms_dec + ms_voc) * 1e-3)},
                  # A8-A10: 4,691,968 FLOP per (padded) mel frame, fp32 MFMA peak 157.3 TFLOP/s; launch-bound at this size
                  "mel_decoder_roofline": {"bound": "mfma(f32)", "achieved": 4_691_968 * B * T_m / (ms_dec * 1e-3) / 1e12,
                                           "peak": 157.3, "unit": "TFLOP/s",
                                           "frac": 4_691_968 * B * T_m / (ms_dec * 1e-3) / 1e12 / 157.3}}
        if gloss_rows is not None and s2pa_ms > 0:
            gbs = 6144.0 * gloss_rows / (s2pa_ms / max(s2pa_n, 1) * 1e-3) / 1e9
            stages["s2pa_roofline"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                                       "kernel": "dtts::s2pa_kernel", "avg_launch_ms": s2pa_ms / max(s2pa_n, 1),
                                       "algorithmic_bytes": "6144 B x unmasked gloss rows (fp32 keys + values, 768 wide)",
                                       "unmasked_gloss_rows": gloss_rows, "padded_gloss_rows": B * T_w * L_k}
    iso = None
    if pipelined and rank == 0:
        # outside the timed region: the same vocoder kernels on the last batch with nothing else on the GPU, so that
        # the kernel family's own rate can be told apart from the rate it reaches while sharing CUs with text->mel
        voc.ctx.timer_reset()
        for _ in range(3):
            voc.forward_batch(nonlocal_mel[0], lens)
        torch.cuda.synchronize()
        iso_ms, iso_launches = voc.ctx.timer_read(abi.TIMER_VOC_CONV)
        iso = (iso_ms, iso_launches, 3 * int(lens.sum().item()))
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        f = torch.tensor([frames_rank], device=dev, dtype=torch.int64)
        dist.all_reduce(f)
        frames_total = int(f.item())
    else:
        frames_total = frames_rank
    assert torch.isfinite(wav).all() and float(wav.abs().max()) <= 1.0

    if args.phases and rank == 0:
        prev_ev = None
        for ev, hs in phase_log[-args.steps:]:
            if prev_ev is not None:
                print("  gap to previous step end (device ms): %.2f" % prev_ev.elapsed_time(ev[0]), file=sys.stderr)
            prev_ev = ev[3]
            print("phases: device ms encode %.2f decode %.2f vocoder %.2f | host ms encode %.2f decode %.2f vocoder %.2f" % (
                ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]),
                (hs[1] - hs[0]) * 1e3, (hs[2] - hs[1]) * 1e3, (hs[3] - hs[2]) * 1e3), file=sys.stderr)
    if rank == 0:
        value = frames_total / elapsed
        samples = value * voc.hop
        achieved = FLOP_PER_FRAME_VOCODER * frames_rank / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # HBM bytes of the same kernel family: PMC counters cannot be read from inside the process, so the committed
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE result of this command (profiles/r*_pmc_traffic.json, corrected
        # as MI355X_MICROARCH.md prescribes) is scaled by this run's frame count; null if the file is absent
        traffic, traffic_src = None, None
        import glob
        tps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))   # the newest round's measurement
        tp = tps[-1] if tps else ""
        if args.precision != "bf16x3" and tp:
            with open(tp) as f:
                tj = json.load(f)
            traffic = tj["hbm_bytes_per_mel_frame"] * (frames_rank / max(args.steps, 1)) / max(conv_launches / max(args.steps, 1), 1)
            traffic_src = f"profiles/{os.path.basename(tp)} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; bytes per launch, avg)"
        out = {
            "metric": "mel-frames/sec, end-to-end text->mel->wav (audio-samples/sec = 256x; RTF reported alongside)",
            "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (random-init weights of the real architecture, Biaobei sentence/dictionary structure)",
            "config": {"workload": "BASELINE configs[1]: Biaobei batch=60 per GPU, full Dict-TTS encoder + FVAE decoder + HifiGAN, "
                                   "predicted durations (~22 frames/char)", "utterances_per_gpu": B, "T_w": T_w, "L_k": L_k,
                       "T_mel_padded": T_mel, "mel_frames_per_step_per_gpu": frames_rank // max(args.steps, 1),
                       "parallelism": f"dp{world}" + ("+allgather(mel)" if gather_state[0] else (f" (allgather off: {gather_state[1]})" if gather_state[1] else "")),
                       "streams": "2 (vocoder of batch i overlaps text->mel of batch i+1)" if pipelined else "1",
                       "acoustic_dtype": "f32 (fp32 MFMA)", "vocoder_dtype": args.precision,
                       "dictionary_input": "resident table + ids" if args.dict_table else "keys/values tensors [B,T_w,L_k,768] (reference API)"},
            "audio_samples_per_sec": samples, "rtf": 22050.0 / samples,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "dtts::vconv_kernel<*> + dtts::vpair_kernel<256|128,*> + dtts::rblock_kernel<*> (every HifiGAN convolution; 30 launches/forward)",
                         "launches": conv_launches,
                         "avg_launch_ms": conv_ms / max(conv_launches, 1), "kernel_ms_per_step": conv_ms / max(args.steps, 1),
                         "algorithmic_flop_per_mel_frame": FLOP_PER_FRAME_VOCODER},
        }
        if stages is not None:
            out["stages"] = stages
        if iso is not None and iso[0] > 0:
            ia = FLOP_PER_FRAME_VOCODER * iso[2] / (iso[0] * 1e-3) / 1e12
            out["roofline"]["note"] = ("achieved/frac are measured inside the timed region, where these kernels share the CUs "
                                       "with the text->mel kernels of the next batch (2 streams); 'isolated' is the same "
                                       "kernels on the last batch run alone right after the timed region")
            out["roofline"]["isolated"] = {"achieved": ia, "frac": ia / PEAK_BF16_TFLOPS, "kernel_ms_per_forward": iso[0] / 3,
                                           "launches": iso[1]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, np, synth)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
