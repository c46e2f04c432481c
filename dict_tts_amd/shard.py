"""Utterance-level data parallelism: one process per GPU, no collective on the data path.

Mirrors the reference's rule for distributing test batches over ranks (tasks/tts/tts_base.py:114-127,148-151):
batches are built for world*max_sentences utterances and rank r keeps elements ``r::world``.  The optional
all-gather of mels (BASELINE.json north_star, "whole-node throughput runs") is the only collective; it works on
fixed-capacity padded buffers plus a length vector, so it is one call regardless of the ragged T_mel per rank.
"""
import torch


def shard_indices(n_items, rank, world, max_sentences=60):
    """-> list of batches (lists of item indices) this rank processes; union over ranks = range(n_items), disjoint"""
    batches = []
    step = world * max_sentences
    for start in range(0, n_items, step):
        chunk = list(range(start, min(start + step, n_items)))
        mine = chunk[rank::world]
        if mine:
            batches.append(mine)
    return batches


def gather_mels(mel, lens, cap, dist, group=None, out=None):
    """mel [B,T,80], lens [B] (any device the backend supports) -> (mel_all [world*B,cap,80], lens_all [world*B]).
    Rank-major order; utterance i of rank r lands at r*B + i.  T may differ per rank (T <= cap)."""
    B, T, C = mel.shape
    assert T <= cap, (T, cap)
    world = dist.get_world_size(group)
    pad = mel.new_zeros(B, cap, C)
    pad[:, :T] = mel
    mel_all = out if out is not None else mel.new_empty(world * B, cap, C)
    lens_all = lens.new_empty(world * B)
    w1 = dist.all_gather_into_tensor(mel_all, pad, group=group, async_op=True)
    w2 = dist.all_gather_into_tensor(lens_all, lens.contiguous(), group=group, async_op=True)
    w1.wait()
    w2.wait()
    return mel_all, lens_all
