"""CPU: the stream / event protocol of artgpu_batch_run_io (art_amd/csrc/artgpu_api.hip io_frame) replayed as a discrete-event model.

Per lane: an upload stream, the context's stream and a download stream; two staging slots on each side and two working images, picked by
the parity of the frame's turn.  The model queues exactly the operations io_frame queues, in its order, with HIP's semantics -- a stream
runs its operations in order; hipStreamWaitEvent waits for the record that was the event's latest WHEN THE WAIT WAS QUEUED -- gives every
operation a random duration, runs the three streams as early as their dependencies allow, and then checks every buffer: a reader of turn i
sees the bytes of turn i (the writer of turn i has finished, the writer of turn i + 2 has not started).  Dropping any one of the protocol's
waits makes the check fail, which the last test shows."""
import random

import pytest


def simulate(nframes, seed, drop=None, pinned_in=True, pinned_out=True):
    rng = random.Random(seed)
    streams = {"U": [], "C": [], "D": []}            # per stream: list of op dicts in program order
    last_record = {}                                   # event name -> op (the record op) as of "now" in host program order
    ops = []

    def op(stream, name, turn, reads=(), writes=(), waits=()):
        o = {"stream": stream, "name": name, "turn": turn, "reads": list(reads), "writes": list(writes),
             "deps": [last_record[e] for e in waits if e in last_record], "dur": rng.uniform(0.1, 3.0)}
        streams[stream].append(o); ops.append(o)
        return o

    def record(stream, ev):
        last_record[ev] = op(stream, "record " + ev, -1)

    for i in range(nframes):
        s = i & 1
        up = "U" if pinned_in else "C"                 # pageable buffers: the copy stays on the context's stream
        down = "D" if pinned_out else "C"
        w = []
        if i >= 2 and drop != "scaled":
            w.append(f"scaled{s}")
        op(up, "h2d", i, writes=[f"in{s}"], waits=w)
        record(up, f"up{s}")
        w = [f"up{s}"] if drop != "up" else []
        if i >= 2 and drop != "down":
            w.append(f"down{s}")                       # the working image / staging-out slot of turn i - 2 has been downloaded
        op("C", "scale_colors", i, reads=[f"in{s}"], writes=["cfa"], waits=w)
        record("C", f"scaled{s}")
        op("C", "pipeline", i, reads=["cfa"], writes=[f"img{s}"])
        op("C", "scanlines", i, reads=[f"img{s}"], writes=[f"out{s}"])
        record("C", f"comp{s}")
        op(down, "d2h", i, reads=[f"out{s}"], waits=[f"comp{s}"] if drop != "comp" else [])
        record(down, f"down{s}")

    # run: each stream in order, an op starts when its predecessor on the stream and its event dependencies have finished
    done = {}
    progress = True
    idx = {k: 0 for k in streams}
    tfree = {k: 0.0 for k in streams}
    while progress:
        progress = False
        for k, lst in streams.items():
            while idx[k] < len(lst):
                o = lst[idx[k]]
                if any(id(d) not in done for d in o["deps"]):
                    break
                start = max([tfree[k]] + [done[id(d)] for d in o["deps"]])
                o["start"], o["end"] = start, start + (0.0 if o["name"].startswith("record") else o["dur"])
                done[id(o)] = o["end"]; tfree[k] = o["end"]; idx[k] += 1; progress = True
    assert all(idx[k] == len(v) for k, v in streams.items()), "deadlock in the model"
    return [o for o in ops if not o["name"].startswith("record")]


def hazards(ops):
    bad = []
    writers = {}
    for o in ops:
        for b in o["writes"]:
            writers.setdefault(b, []).append(o)
    for o in ops:
        for b in o["reads"]:
            ws = sorted(writers[b], key=lambda w: w["turn"])
            mine = [w for w in ws if w["turn"] == o["turn"]][0]
            if mine["end"] > o["start"] + 1e-12:
                bad.append((o["name"], o["turn"], b, "reads before its writer finished"))
            later = [w for w in ws if w["turn"] > o["turn"]]
            if later and later[0]["start"] < o["end"] - 1e-12:
                bad.append((o["name"], o["turn"], b, f"overwritten by turn {later[0]['turn']} while being read"))
    return bad


@pytest.mark.parametrize("pinned_in,pinned_out", [(True, True), (False, True), (True, False), (False, False)])
def test_protocol_has_no_hazard(pinned_in, pinned_out):
    for seed in range(200):
        ops = simulate(9, seed, pinned_in=pinned_in, pinned_out=pinned_out)
        assert not hazards(ops), (seed, hazards(ops)[:3])


def test_copies_really_overlap_the_kernels():
    """the point of the three streams: in steady state an upload and a download run while the context's stream computes"""
    ops = simulate(12, 3)
    c = [(o["start"], o["end"]) for o in ops if o["stream"] == "C"]
    def overlaps(o):
        return any(s < o["end"] and o["start"] < e for s, e in c)
    assert sum(overlaps(o) for o in ops if o["name"] == "h2d" and o["turn"] >= 2) >= 6
    assert sum(overlaps(o) for o in ops if o["name"] == "d2h" and o["turn"] < 10) >= 6


@pytest.mark.parametrize("drop", ["scaled", "up", "down", "comp"])
def test_every_wait_of_the_protocol_is_needed(drop):
    assert any(hazards(simulate(9, seed, drop=drop)) for seed in range(200)), drop
