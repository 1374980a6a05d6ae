"""CPU, world_size 2, gloo: the N>1 path of bench.py (frame sharding + completion all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from art_amd import batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = batch.frames_for_rank(nframes, rank, world)
    # "process" each frame: checksum of (frame id, a frame-dependent word)
    cs = batch.checksum64([w for f in mine for w in (f, f * 2654435761 + 12345)])
    recs, tmax = batch.complete_batch(dist, torch.device("cpu"), rank, len(mine), 0, cs, 0.001 * (rank + 1))
    q.put((rank, mine, recs, tmax))
    dist.destroy_process_group()


def _worker_rccl_fallback(rank, world, port, q):
    """open_rccl with a context that cannot do the RCCL gather on one rank only: the MIN-reduce has to leave BOTH ranks without a
    communicator, i.e. on the torch.distributed path (a one-sided fallback would deadlock)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Ctx:          # rank 0 "has" the entry point, rank 1 does not
        pass
    ctx = Ctx()
    if rank == 0:
        ctx.batch_complete = lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not be called"))
    handle = batch.open_rccl(ctx, dist, torch.device("cpu"), rank, world)
    assert handle is None
    recs, tmax, via = batch.complete_batch_rccl(ctx, handle, dist, torch.device("cpu"), rank, world, 3 + rank, 0, 1000 + rank, 0.001 * (rank + 1))
    q.put((rank, recs, tmax, via))
    dist.destroy_process_group()


def test_rccl_completion_falls_back_on_all_ranks_or_none():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rccl_fallback, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, recs, tmax, via in res:
        assert via == "torch.distributed"
        assert [r["frames"] for r in recs] == [3, 4] and [r["checksum"] for r in recs] == [1000, 1001]
        assert abs(tmax - 0.002) < 1e-6


def test_frames_partition_round_robin():
    for world in (1, 2, 4, 8):
        owned = [batch.frames_for_rank(8, r, world) for r in range(world)]
        assert sorted(f for o in owned for f in o) == list(range(8))
        assert all(len(o) == 8 // world for o in owned)
    with pytest.raises(ValueError):
        batch.frames_for_rank(8, 2, 2)


def test_two_rank_completion_gather():
    world, nframes = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for rank, mine, recs, tmax in res:
        assert [r["rank"] for r in recs] == [0, 1]
        assert [r["frames"] for r in recs] == [3, 2]
        assert all(r["status"] == 0 for r in recs)
        assert abs(tmax - 0.002) < 1e-6          # MAX over ranks
    # every rank sees the same records, and they match a local recomputation
    assert res[0][2] == res[1][2]
    exp0 = batch.checksum64([w for f in (0, 2, 4) for w in (f, f * 2654435761 + 12345)])
    assert res[0][2][0]["checksum"] == exp0


def test_bench_py_two_rank_path_dry():
    """`python bench.py --gpus 2` has to start its own two ranks (no launcher, no WORLD_SIZE) and print ONE line with n_gpus 2 and two
    completion records; --dry swaps RCCL for gloo and the device work for a no-op, everything else is bench.py's own N>1 code path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry", "--steps", "3", "--warmup", "1",
                        "--width", "256", "--height", "192"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry"] is True and d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["completion_records"] == 2 and d["config"]["frames_per_step"] == 2
    assert d["config"]["completion_via"] == "torch.distributed (gloo, dry run)"
    recs = d["config"]["records"]
    assert [x["rank"] for x in recs] == [0, 1] and all(x["frames"] == 3 and x["status"] == 0 for x in recs)
    assert recs[0]["checksum"] != recs[1]["checksum"]        # rank r works on frame r of the batch (different seeds)


@pytest.mark.parametrize("gpus,workload", [(8, "c3"), (2, "c4")])
def test_bench_py_rank_path_dry_world8_and_config4(gpus, workload):
    """BASELINE configs[3] is eight ranks of the c4 per-frame pipe: the launcherless start, the sharding and the completion gather of
    `bench.py --gpus 8` (and of `--workload c4`) on the CPU -- gloo instead of RCCL, no device work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--dry", "--workload", workload, "--steps", "2", "--warmup", "1",
                        "--width", "256", "--height", "192"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry"] is True and d["n_gpus"] == gpus and d["scaling"] == "weak"
    c = d["config"]
    assert c["completion_records"] == gpus and c["frames_per_step"] == gpus and c["completion_via"] == "torch.distributed (gloo, dry run)"
    assert c["workload_flag"] == workload
    assert [x["rank"] for x in c["records"]] == list(range(gpus)) and all(x["frames"] == 2 and x["status"] == 0 for x in c["records"])
    assert len({x["checksum"] for x in c["records"]}) == gpus          # rank r works on frame r of the batch


def _worker_failing(rank, world, port, q):
    """rank 1's device work failed (status 7 after one frame): it still takes part in the completion gather, so that rank 0 does not hang"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    status, frames = (7, 1) if rank == 1 else (0, 3)
    recs, tmax = batch.complete_batch(dist, torch.device("cpu"), rank, frames, status, 100 + rank, 0.001 * (rank + 1))
    q.put((rank, recs, tmax))
    dist.destroy_process_group()


def test_failing_rank_still_completes_the_gather_on_all_ranks():
    """The reference's queue is one process and stops at the job that failed (simpleprocess.cc:600-602).  Here every rank is a process of its
    own: a rank whose frame failed reports status != 0 and the frames it finished, and EVERY rank gets all the records -- nobody is left
    waiting in the collective."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_failing, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                      # both ranks hold the same two records
    recs = res[0][1]
    assert [r["status"] for r in recs] == [0, 7] and [r["frames"] for r in recs] == [3, 1]


@pytest.mark.parametrize("gpus,bad", [(2, 1), (4, 0)])
def test_bench_py_failing_rank_dry(gpus, bad):
    """bench.py's own N > 1 path with one rank whose second timed step raises (--dry-fail-rank): the failing rank skips its remaining
    steps but joins every barrier and the completion gather (StepGuard), rank 0 prints ONE line that names the failed rank and counts only
    the other ranks' frames, and the job's exit code is 3 -- no rank hangs (the timeout would catch it)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--dry", "--steps", "3", "--warmup", "1",
                        "--width", "256", "--height", "192", "--dry-fail-rank", str(bad)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["failed_ranks"] == [bad] and d["n_gpus"] == gpus
    recs = d["config"]["records"]
    assert d["config"]["completion_records"] == gpus and [x["rank"] for x in recs] == list(range(gpus))
    for x in recs:
        if x["rank"] == bad:
            assert x["status"] != 0 and x["frames"] == 1      # the first timed step finished, the second raised
        else:
            assert x["status"] == 0 and x["frames"] == 3
    assert f"[bench rank {bad}] step failed" in r.stderr
