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


def _records_to_dicts(rows):
    out = []
    for v in rows:
        out.append({"rank": int(v[0]), "frames": int(v[1]), "status": int(v[2]), "checksum": int(v[3]) & 0xFFFFFFFFFFFFFFFF, "elapsed_us": int(v[4])})
    return out, max(o["elapsed_us"] for o in out) / 1e6


class RcclComm:
    """An RCCL communicator created through librccl's C API (what a non-Python host would do); `comm` is the ncclComm_t."""

    def __init__(self, lib, comm):
        self.lib, self.comm = lib, comm

    def close(self):
        import ctypes as C
        if self.comm is not None:
            self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


def open_rccl(ctx, dist, device, rank: int, world: int, group=None):
    """Create the communicator `artgpu_batch_complete` gathers over -- BEFORE the timed region: communicator set-up takes seconds and is
    not part of the batch.  The unique id travels over the already initialised torch.distributed group.  Every step that can fail on one
    rank only (loading librccl, ncclGetUniqueId on rank 0, ncclCommInitRank) is followed by a MIN-reduce of "did it work here", so that
    either ALL ranks get a communicator or ALL of them get None and complete through torch.distributed: a rank that raised or fell back
    on its own would leave the others hanging in the next collective.

    `group`: the process group the set-up's own collectives (the agreements, the broadcast of the id) run on.  A caller that runs this
    under a watchdog passes a group of its own (dist.new_group(), created beforehand by every rank): should the set-up hang and the caller
    give up on it, a rank may still be parked inside one of these collectives -- on a dedicated group that cannot pair up with the
    collectives the caller goes on to issue on the default group."""
    import ctypes as C
    import torch

    def agree(flag: bool) -> bool:
        ok = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        return int(ok.item()) == 1

    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = C.CDLL(name)
            break
        except OSError:
            continue
    if not agree(rccl is not None and hasattr(ctx, "batch_complete")):
        return None

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    got_id = rank != 0 or rccl.ncclGetUniqueId(C.byref(uid)) == 0
    if not agree(got_id):
        return None
    wire = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=device)
    if world > 1:
        dist.broadcast(wire, src=0, group=group)
    C.memmove(C.byref(uid), bytes(wire.cpu().tolist()), 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(device)
    inited = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    handle = RcclComm(rccl, comm if inited else None)
    if not agree(inited):
        handle.close()
        return None
    return handle


def complete_batch_rccl(ctx, handle, dist, device, rank: int, world: int, frames_done: int, status: int, checksum: int, elapsed_s: float):
    """The completion step through the library's own entry point: `artgpu_batch_complete` all-gathers the records over the communicator
    of open_rccl().  handle None (no communicator, on every rank by construction): complete_batch() over torch.distributed.
    Returns (records, max_elapsed_s, "rccl-capi" | "torch.distributed")."""
    if handle is None:
        recs, el = complete_batch(dist if world > 1 else None, device, rank, frames_done, status, checksum, elapsed_s)
        return recs, el, "torch.distributed"
    cs = checksum - (1 << 64) if checksum >= (1 << 63) else checksum
    rows = ctx.batch_complete([rank, frames_done, status, cs, int(elapsed_s * 1e6), 0, 0, 0], nranks=world, rccl_comm=handle.comm)
    recs, el = _records_to_dicts(rows)
    return recs, el, "rccl-capi"
