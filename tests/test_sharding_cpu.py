"""world_size-2 gloo test (CPU) of the N>1 path: the reference's rank::world utterance sharding and the padded
all-gather of mels with ragged T_mel per rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dict_tts_amd.shard import exchange_shapes, gather_mels, group_device, n_steps, ranks_seen, shard_indices


def test_shard_indices_partition():
    for n, world, ms in ((200, 8, 60), (200, 2, 60), (7, 4, 2), (60, 1, 60)):
        seen = []
        for r in range(world):
            for b in shard_indices(n, r, world, ms):
                assert len(b) <= ms
                seen += b
        assert sorted(seen) == list(range(n))
    # the reference's rule: element i of a world*max_sentences chunk goes to rank i % world
    assert shard_indices(10, 1, 2, 3) == [[1, 3, 5], [7, 9]]
    # BASELINE configs[2]: the 200-sentence test set over 8 GPUs = one chunk, utterance i -> rank i mod 8, 25 each
    assert all(shard_indices(200, r, 8, 60) == [list(range(r, 200, 8))] for r in range(8)) and n_steps(200, 8, 60) == 1
    # the tail chunk may leave ranks without a batch: the step count is common, the batch lists are not
    assert n_steps(2, 4, 1) == 1 and [len(shard_indices(2, r, 4, 1)) for r in range(4)] == [1, 1, 0, 0]
    assert n_steps(200, 1, 60) == 4 and [len(b) for b in shard_indices(200, 0, 1, 60)] == [60, 60, 60, 20]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # step 0: ragged B and T_mel per rank (rank 0: 3 x 8, rank 1: 2 x 5); step 1: rank 1 has NO batch (the tail chunk of
    # shard_indices) but still enters the collectives; step 2: nobody has one
    shapes = {0: {0: (3, 8), 1: (2, 5)}, 1: {0: (1, 4), 1: None}, 2: {0: None, 1: None}}

    def mk(r, step):
        bt = shapes[step][r]
        if bt is None:
            return None, None
        B, T = bt
        mel = torch.arange(B * T * 80, dtype=torch.float32).reshape(B, T, 80) + 1000 * r + 7 * step
        return mel, torch.arange(T, T - B, -1, dtype=torch.int32)

    for step in range(3):
        mel, lens = mk(rank, step)
        mel_all, lens_all, meta = gather_mels(mel, lens, dist)
        want_meta = [list(shapes[step][r] or (0, 0)) for r in range(world)]
        ok = ok and meta.tolist() == want_meta
        if step == 2:
            ok = ok and mel_all is None and lens_all is None
            continue
        Bm, Tm = max(m[0] for m in want_meta), max(m[1] for m in want_meta)
        ok = ok and tuple(mel_all.shape) == (world, Bm, Tm, 80) and tuple(lens_all.shape) == (world, Bm)
        for r in range(world):
            wm, wl = mk(r, step)
            if wm is None:
                ok = ok and float(mel_all[r].abs().max()) == 0 and lens_all[r].tolist() == [0] * Bm
                continue
            B, T = wm.shape[:2]
            ok = ok and torch.equal(mel_all[r, :B, :T], wm) and float(mel_all[r, B:].abs().sum()) == 0
            ok = ok and float(mel_all[r, :, T:].abs().sum()) == 0 and lens_all[r, :B].tolist() == wl.tolist()
    # round 6: (a) the shape exchange started EARLY (right behind encode, where the host already knows B and T_mel) and read only inside
    # gather_mels: same results, two exchanges of consecutive steps may be in flight at once; (b) the fixed-capacity gather, with no shape
    # exchange at all (meta None; lens_all says which utterances exist); a batch beyond the capacity raises on the rank that holds it
    early = [exchange_shapes(*(shapes[step][rank] or (0, 0)), dist) for step in (0, 1)]
    for step in (0, 1):
        mel, lens = mk(rank, step)
        mel_all, lens_all, meta = gather_mels(mel, lens, dist, shapes=early[step])
        mel_ref, lens_ref, meta_ref = gather_mels(mel, lens, dist)
        ok = ok and torch.equal(mel_all, mel_ref) and torch.equal(lens_all, lens_ref) and torch.equal(meta, meta_ref)
        mel_cap, lens_cap, meta_cap = gather_mels(mel, lens, dist, capacity=(4, 9))
        ok = ok and meta_cap is None and tuple(mel_cap.shape) == (world, 4, 9, 80) and tuple(lens_cap.shape) == (world, 4)
        Bm, Tm = mel_ref.shape[1:3]
        ok = ok and torch.equal(mel_cap[:, :Bm, :Tm], mel_ref) and float(mel_cap[:, Bm:].abs().sum()) == 0 and float(mel_cap[:, :, Tm:].abs().sum()) == 0
        ok = ok and torch.equal(lens_cap[:, :Bm], lens_ref) and int(lens_cap[:, Bm:].abs().sum()) == 0
    try:
        gather_mels(torch.zeros(5, 3, 80), torch.ones(5, dtype=torch.int32), dist, capacity=(4, 9))   # (raises BEFORE any collective: both ranks stay matched)
        ok = False
    except ValueError:
        pass
    # the collectives' device follows the BACKEND (a rank without a batch must not fall back to a different device kind)
    ok = ok and group_device(dist) == torch.device("cpu")
    seen = ranks_seen(dist, device_index=10 + rank, device_uuid=f"GPU-fake-{rank}")
    ok = ok and seen == [{"rank": r, "device_index": 10 + r, "device_uuid": f"GPU-fake-{r}"} for r in range(world)]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_mels_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_group_device_follows_backend():
    """with an nccl (RCCL) group a rank WITHOUT a batch must create its collective buffers on its CUDA device, not the CPU."""
    class FakeDist:
        def __init__(self, b):
            self.b = b

        def get_backend(self, group=None):
            return self.b

    import unittest.mock as mock
    with mock.patch("torch.cuda.current_device", return_value=3):
        assert group_device(FakeDist("nccl")) == torch.device("cuda", 3)
        assert group_device(FakeDist("cpu:gloo,cuda:nccl")) == torch.device("cuda", 3)
    assert group_device(FakeDist("gloo")) == torch.device("cpu")
