"""world_size-2 gloo test (CPU) of the N>1 path: the reference's rank::world utterance sharding and the padded
all-gather of mels with ragged T_mel per rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dict_tts_amd.shard import gather_mels, shard_indices


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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, cap = 3, 12
    T = 8 if rank == 0 else 5            # ragged T_mel per rank
    mel = torch.arange(B * T * 80, dtype=torch.float32).reshape(B, T, 80) + 1000 * rank
    lens = torch.tensor([T, T - 1, T - 2], dtype=torch.int32)
    mel_all, lens_all = gather_mels(mel, lens, cap, dist)
    ok = mel_all.shape == (world * B, cap, 80)
    for r in range(world):
        Tr = 8 if r == 0 else 5
        want = torch.arange(B * Tr * 80, dtype=torch.float32).reshape(B, Tr, 80) + 1000 * r
        ok = ok and torch.equal(mel_all[r * B:(r + 1) * B, :Tr], want) and float(mel_all[r * B:(r + 1) * B, Tr:].abs().max()) == 0
        ok = ok and lens_all[r * B:(r + 1) * B].tolist() == [Tr, Tr - 1, Tr - 2]
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
