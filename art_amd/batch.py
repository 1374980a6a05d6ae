"""Batch sharding: one frame per GPU, no data-path collective.

The reference's batch queue processes frames strictly one after another
(rtengine/simpleprocess.cc:591-611 `batchProcessingThread`); frames share no state, so a batch
is sharded frame *i* -> rank *i mod world*.  The only collective is the completion step: an
all-gather of one fixed-size record per rank (frames done, status, 64-bit checksum, elapsed
microseconds), which doubles as the barrier, plus a MAX-reduce of the elapsed time for
``bench.py``.  Works over RCCL (backend "nccl") on GPUs and gloo on CPU.
"""
from __future__ import annotations

from typing import List, Sequence

RECORD_WORDS = 8  # 64-byte completion record: int64[8]


def frames_for_rank(nframes: int, rank: int, world: int) -> List[int]:
    """Frame indices owned by `rank` (round-robin, the reference's queue order within a rank)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, nframes, world))


def checksum64(words: Sequence[int]) -> int:
    """Order-sensitive 64-bit FNV-1a over 64-bit words (used for the completion record)."""
    h = 0xCBF29CE484222325
    for w in words:
        h ^= int(w) & 0xFFFFFFFFFFFFFFFF
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def complete_batch(dist, device, rank: int, frames_done: int, status: int, checksum: int, elapsed_s: float):
    """All-gather the per-rank completion records; returns (records[list of dict], max_elapsed_s).

    `dist` is torch.distributed (initialised) or None for a single process."""
    import torch

    rec = torch.zeros(RECORD_WORDS, dtype=torch.int64, device=device)
    rec[0] = rank
    rec[1] = frames_done
    rec[2] = status
    rec[3] = checksum - (1 << 64) if checksum >= (1 << 63) else checksum  # two's complement into int64
    rec[4] = int(elapsed_s * 1e6)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        allrec = [rec]
    else:
        allrec = [torch.zeros_like(rec) for _ in range(dist.get_world_size())]
        dist.all_gather(allrec, rec)
    out = []
    for r in allrec:
        v = [int(x) for x in r.cpu().tolist()]
        out.append({"rank": v[0], "frames": v[1], "status": v[2], "checksum": v[3] & 0xFFFFFFFFFFFFFFFF, "elapsed_us": v[4]})
    return out, max(o["elapsed_us"] for o in out) / 1e6
