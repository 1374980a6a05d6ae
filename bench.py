#!/usr/bin/env python3
"""bench.py -- megapixels/second of the raw-development hot path on MI355X.

A "step" is one pass of the hot path over one synthetic 45 MP (8192x5464) Bayer frame per
GPU, CFA already resident in HBM, result left resident in HBM.
Frames are independent, so N GPUs process N frames per step with no data-path collective
(weak scaling); the only collective is the completion barrier / max-over-ranks of the time.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : dominant kernel's algorithmic bytes / its average launch time vs 8 TB/s HBM
  cpu_baseline : the CPU oracle (port of the reference's x86-64 path) timed on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W45, H45 = 8192, 5464
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_PX = 16         # 4 B CFA in + 12 B RGB out
BARRIER_AWARE_B_PER_PX = {"amaze": 16, "rcd": 16, "c3": 16 + 760, "c4": 16 + 760 + 12 + 40, "c5": 16 + 760}     # SURVEY.md section 8d, last row


class StepGuard:
    """A step that fails on ONE rank must not strand the others: the reference's queue stops at the failing job (simpleprocess.cc:600-602,
    one process, one queue); here the ranks are independent processes that meet at the barriers and at the completion gather, so a rank whose
    step raised keeps quiet for the rest of the run (no further device work), still joins every barrier and the gather, and reports
    status != 0 and the steps it completed in its 64-byte record.  Every rank then sees the failure in the gathered records, rank 0 prints a
    line with `failed_ranks` (value = the frames of the ranks that succeeded), and all ranks exit with code 3."""

    def __init__(self, rank):
        self.rank, self.status, self.done = rank, 0, 0

    def run(self, fn, *a):
        if self.status:
            return None
        try:
            r = fn(*a)
            self.done += 1
            return r
        except Exception as e:      # noqa: BLE001
            msg = str(e)
            self.status = int(msg[1:msg.index("]")]) if msg.startswith("[") and "]" in msg and msg[1:msg.index("]")].lstrip("-").isdigit() else 1
            self.status = self.status or 1
            print(f"[bench rank {self.rank}] step failed, status {self.status}: {msg}", file=sys.stderr, flush=True)
            return None

    def done_timed(self, warmup):
        return max(0, self.done - warmup)


def failure_line(records, failed, elapsed, mp, args, world, dry=False):
    good = sum(r["frames"] for r in records if r["status"] == 0)
    return {"metric": "megapixels/sec end-to-end (AMaZE+FTblockDN+tone), 45 MP Bayer", "dry": dry, "failed_ranks": failed,
            "value": round(args.lanes * good * mp / max(elapsed, 1e-9), 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "data": "synthetic",
            "config": {"workload": "INCOMPLETE: a rank's step failed; value counts the frames of the ranks that succeeded", "workload_flag": args.workload,
                       "completion_records": len(records), "records": records}}


def dry_main(args, rank, world, dist, torch, synth) -> None:
    """The rank-side control flow of a real run -- frame `rank` of the batch, warm-up, barrier, K timed steps, barrier, completion
    all-gather with MAX of the elapsed time, one JSON line from rank 0 -- with gloo and without a device (no kernel is called)."""
    dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from art_amd import batch
    W, H = args.width, args.height
    raw = synth.bayer_frame(W, H, synth.FILTERS_RGGB, seed=rank)
    mine = batch.frames_for_rank(world, rank, world)            # one frame per rank per step
    assert mine == [rank]

    def barrier():
        if world > 1:
            dist.barrier()

    def step(k=-1):
        if args.dry_fail_rank == rank and k == min(1, args.steps - 1):
            raise RuntimeError("simulated device failure (--dry-fail-rank)")
        return int(raw[H // 2, W // 2])                          # stands in for the device work on this rank's frame

    guard = StepGuard(rank)
    for _ in range(args.warmup):
        guard.run(step)
    barrier()
    t0 = time.perf_counter()
    probe = 0
    for k in range(args.steps):
        r_ = guard.run(step, k)
        probe = probe if r_ is None else r_
    barrier()
    t1 = time.perf_counter()
    records, elapsed = batch.complete_batch(dist if world > 1 else None, dev, rank, guard.done_timed(args.warmup), guard.status, batch.checksum64([probe]), t1 - t0)
    failed = [r["rank"] for r in records if r["status"] != 0]
    if failed:
        if rank == 0:
            print(json.dumps(failure_line(records, failed, elapsed, W * H / 1e6, args, world, dry=True)), flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"metric": "megapixels/sec end-to-end (AMaZE+FTblockDN+tone), 45 MP Bayer", "dry": True,
                          "value": round(world * args.lanes * args.steps * W * H / 1e6 / max(elapsed, 1e-9), 2), "unit": "MP/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak", "data": "synthetic",
                          "config": {"workload": "dry run (no device work)", "workload_flag": args.workload, "frames_per_step": world * args.lanes,
                                     "parallelism": f"frame-per-gpu x{world}", "completion_records": len(records),
                                     "completion_via": "torch.distributed (gloo, dry run)" if world > 1 else "single process",
                                     "records": records}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=W45)
    ap.add_argument("--height", type=int, default=H45)
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)      # --dry only: this rank's second timed step raises
    ap.add_argument("--cpu-sample-only", action="store_true", help="cpu_baseline: only the bounded region sample, not the whole frame once")
    ap.add_argument("--xtrans-passes", type=int, default=3, choices=[1, 3],
                    help="c5: the Markesteijn method BASELINE.md C5 names -- 3 = THREE_PASS (CIELab homogeneity, xtrans_demosaic.cc:477-651), "
                         "1 = ONE_PASS (YPbPr, L688-741); the default line is 3-pass and carries the 1-pass step as `xtrans_one_pass`")
    ap.add_argument("--workload", default="c3", choices=["amaze", "rcd", "c3", "c4", "c5"],
                    help="amaze/rcd: demosaic only (BASELINE configs[1]); c3: AMaZE + getImage/matrix + FTblockDN wavelet "
                         "denoise + exposure + tone curve (BASELINE configs[2], the configuration the metric is quoted on); "
                         "c4: c3 + guided chroma smoothing + NL-means (the per-frame pipe of BASELINE configs[3]); "
                         "c5: X-Trans 3-pass (Markesteijn, CIELab) + the c3 stages on a 100 MP 11648x8736 frame (BASELINE configs[4])")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="only the timed steps (no NEUTRAL / separate-stage / two-frames legs behind them): kernel traces and counter passes")
    ap.add_argument("--separate-stages", action="store_true",
                    help="one library call per stage of the reference's order (getImage, denoise, exposure) instead of the fused tool "
                         "artgpu_improc_denoise_fused; same result, the image written and read again between the stages")
    ap.add_argument("--lanes", type=int, default=1,
                    help="frames in flight per GPU (each on its own context, stream and host thread; a step is then `lanes` frames). "
                         "Default 1 = BASELINE's one frame per GPU; 3 gives +11 %% throughput at 45 MP (independent frames fill each "
                         "other's latency gaps).  Pipeline workloads only.")
    ap.add_argument("--tone", default="std", choices=["std", "neutral"],
                    help="tone-curve mode of the last stage: STD (3 LUT lookups per pixel; the headline line) or NEUTRAL, ART's default mode "
                         "(Jzazbz hue-preserving curve, curves.cc:854-1038)")
    ap.add_argument("--dry", action="store_true",
                    help="run the rank / sharding / completion code path without a GPU: gloo instead of RCCL, every device call replaced by a "
                         "no-op (tests/test_multiproc.py drives `bench.py --gpus 2 --dry` on the CPU); the printed line carries \"dry\": true "
                         "and is not a measurement")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="artgpu_set_option on every context of the run (A/B switches, e.g. dn_streams=0)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="after the timed K steps: back-to-back steps for at least this long (the batch queue of simpleprocess.cc:591-611 runs "
                         "for minutes, the contract region for a fraction of a second at boost clocks); reported as `sustained`.  0: off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same command the driver
    # uses: torch.distributed.run on 127.0.0.1) and hand over to them.  Under a launcher WORLD_SIZE is set and we are one rank.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import torch.distributed as dist

    from art_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry:
        dry_main(args, rank, world, dist, torch, synth)
        return
    from art_amd import capi
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H = args.width, args.height
    xtrans = args.workload == "c5"
    xt_passes = args.xtrans_passes
    if xtrans and (W, H) == (W45, H45):
        W, H = 11648, 8736                                    # 100 MP X-Trans sensor
    filt = synth.FILTERS_RGGB
    raw = synth.xtrans_frame(W, H, seed=rank) if xtrans else synth.bayer_frame(W, H, filt, seed=rank)   # frame `rank` of the batch
    d_raw = torch.from_numpy(raw).to(dev)
    d_out = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(3)]
    stream = torch.cuda.current_stream(dev)
    def apply_opts(c):
        for o in args.opt:
            name, _, val = o.partition("=")
            c.set_option(name, int(val, 0))
        return c

    ctx = apply_opts(capi.Context(local_rank, stream.cuda_stream))
    out = capi.RGB(*[capi.device_plane(t) for t in d_out])
    p_raw = capi.device_plane(d_raw)
    method = capi.BAYER_RCD if args.workload == "rcd" else capi.BAYER_AMAZE
    border = 7 if xtrans else 4
    iw, ih = W - 2 * border, H - 2 * border
    pipeline = args.workload in ("c3", "c4", "c5")
    smoothing = args.workload == "c4"
    if pipeline:
        d_img = [torch.empty((ih, iw), dtype=torch.float32, device=dev) for _ in range(3)]
        img = capi.RGB(*[capi.device_plane(t) for t in d_img])
        mul = (2.1374, 1.0, 1.5918)                            # rm, gm, bm of a daylight WB
        mat = np.array([[0.6325, 0.2312, 0.0921], [0.2198, 0.7712, 0.0090], [0.0166, 0.0713, 0.7514]])  # raw -> Rec2020
        ws = np.array([[0.6734241, 0.1656411, 0.1251286], [0.2790177, 0.6753402, 0.0456377], [-0.0019300, 0.0299784, 0.7973330]])  # Rec2020
        iws_n = np.array([[1.6473376, -0.3935675, -0.2359961], [-0.6826036, 1.6475887, 0.0128190], [0.0296524, -0.0628993, 1.2531279]])  # its inverse (iccmatrices.h:157-161)
        dn = capi.DenoiseToolParams(capi.DenoiseParams(40.0, 50.0, 0, 15.0, 0.0, 0.0, 1.7, 0, 0, 0), 1 if smoothing else 0, 3,
                                    50 if smoothing else 0, 80)
        ccurve, _ = capi.noise_curve_lut()                     # the chroma noise curve ImProcFunctions::denoise always installs
        expcomp = 0.3
        x = np.arange(65536, dtype=np.float64) / 65535.0
        lut = ((1.0 - np.cos(np.pi * x ** 0.7)) / 2.0 * 65535.0).astype(np.float32)   # fixed S-curve (an input of the stage)
        exp_scale = float(np.float32(2.0 ** expcomp))
        # default: getImage + the camera matrix in front of ImProcFunctions::denoise and the STAGE_1 exposure behind it run inside the tool's
        # own first / last pixel pass (artgpu_improc_denoise_fused: the same operations per pixel, the image not written and read again
        # between the stages); --separate-stages: one call per stage of the reference's order
        fused_tool = not args.separate_stages
        stage_names = ["demosaic", "get_image+matrix+denoise+exposure", "tone_curve"] if fused_tool else ["demosaic", "get_image+matrix", "denoise", "exposure", "tone_curve"]
    else:
        stage_names = ["demosaic"]
    stage_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(stage_names) + 1)] for _ in range(args.steps)]
    cur = {"ev": None}

    def mark(k):
        if cur["ev"] is not None:
            cur["ev"][k].record(stream)

    # --lanes: additional frames in flight, each with its own context / stream / buffers / frame of the batch
    import threading
    extra = []

    def make_lane(k):
        st = torch.cuda.Stream(dev)
        lr = synth.xtrans_frame(W, H, seed=rank + 1000 * k) if xtrans else synth.bayer_frame(W, H, filt, seed=rank + 1000 * k)
        ln = {"ctx": apply_opts(capi.Context(local_rank, st.cuda_stream)), "stream": st, "raw": torch.from_numpy(lr).to(dev),
              "out": [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(3)],
              "img": [torch.empty((ih, iw), dtype=torch.float32, device=dev) for _ in range(3)]}
        ln["p_raw"] = capi.device_plane(ln["raw"]); ln["p_out"] = capi.RGB(*[capi.device_plane(t) for t in ln["out"]])
        ln["p_img"] = capi.RGB(*[capi.device_plane(t) for t in ln["img"]])
        return ln

    # With frames in flight the issue-bound AMaZE stream kernel leaves 3/8 of the CUs to the other frames' memory-bound passes
    # (artgpu_batch_run does this for its lanes by itself; the lanes here are separate contexts, so they are told: option amaze_grid)
    user_grid = any(o.startswith("amaze_grid=") for o in args.opt)
    inflight_grid = max(1, torch.cuda.get_device_properties(dev).multi_processor_count * 5 // 8)

    def in_flight(c, on):
        if not user_grid and not xtrans:
            c.set_option("amaze_grid", inflight_grid if on else 0)

    if args.lanes > 1:
        if not pipeline:
            raise SystemExit("--lanes needs a pipeline workload (c3/c4/c5)")
        in_flight(ctx, True)
        for k in range(1, args.lanes):
            extra.append(make_lane(k))
            in_flight(extra[-1]["ctx"], True)

    lane_errors = []

    def lane_frame(ln):
        # (runs on its own host thread: an exception there must stop the bench, not vanish with the thread)
        try:
            lane_frame_body(ln)
        except BaseException as e:      # noqa: BLE001
            lane_errors.append(e)

    def lane_frame_body(ln):
        c = ln["ctx"]
        if xtrans:
            c.demosaic_xtrans(xt_passes, xt_passes == 3, ln["p_raw"], synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, ln["p_out"])
        else:
            c.demosaic_bayer(method, ln["p_raw"], filt, 1.0, border, ln["p_out"])
        if fused_tool:
            c.improc_denoise_fused(ln["p_img"], dn, ws, demosaiced=ln["p_out"], sx1=border, sy1=border, mul=mul, do_clip=True, cam_to_work=mat,
                                   exposure=(exp_scale, 0.0), ecomp=expcomp, calclum_mat=mat, noise_c_curve=ccurve)
        else:
            c.get_image(ln["p_out"], border, border, mul, True, mat, ln["p_img"])
            c.improc_denoise(ln["p_img"], dn, ws, ecomp=expcomp, calclum_mat=mat, noise_c_curve=ccurve)
            c.exposure(ln["p_img"], exp_scale, 0.0)
        c.tone_curve(ln["p_img"], lut, 1.0, True)

    def step():
        if extra:
            th = [threading.Thread(target=lane_frame, args=(ln,)) for ln in extra]
            for t_ in th:
                t_.start()
            step0()
            for t_ in th:
                t_.join()
            if lane_errors:
                raise lane_errors[0]
        else:
            step0()

    def step0():
        mark(0)
        if xtrans:
            ctx.demosaic_xtrans(xt_passes, xt_passes == 3, p_raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, out)
        else:
            ctx.demosaic_bayer(method, p_raw, filt, 1.0, border, out)
        mark(1)
        if pipeline:
            # ImProcFunctions::denoise: ccalc map, expcomp(+), RGB_denoise (shrinkage + DCT detail recovery),
            # [guided smoothing, NL-means], expcomp(-)
            if fused_tool:
                ctx.improc_denoise_fused(img, dn, ws, demosaiced=out, sx1=border, sy1=border, mul=mul, do_clip=True, cam_to_work=mat,
                                         exposure=(exp_scale, 0.0), ecomp=expcomp, calclum_mat=mat, noise_c_curve=ccurve)
                mark(2)
            else:
                ctx.get_image(out, border, border, mul, True, mat, img)
                mark(2)
                ctx.improc_denoise(img, dn, ws, ecomp=expcomp, calclum_mat=mat, noise_c_curve=ccurve)
                mark(3)
                ctx.exposure(img, exp_scale, 0.0)
                mark(4)
            if args.tone == "neutral":
                ctx.tone_curve_neutral(img, lut, 1.0, ws, iws_n)
            else:
                ctx.tone_curve(img, lut, 1.0, True)
            mark(len(stage_names))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The completion gather's RCCL communicator (artgpu_batch_complete, the C-ABI form of the batch's one collective) is opened here,
    # before the warm-up: its set-up takes seconds and belongs to the job's start, not to the timed steps or to what follows them.
    from art_amd import batch
    use_rccl_capi = "WORLD_SIZE" in os.environ and not os.environ.get("ARTGPU_BENCH_TORCH_GATHER")
    rccl_handle, rccl_stuck = None, False
    if use_rccl_capi:
        # (in a watchdog thread: communicator set-up is a rendezvous of all ranks -- should it hang on some node, every rank times out the
        # same way, the completion step goes through torch.distributed and the process leaves with os._exit past the stuck thread)
        # The set-up's own collectives (agreements, the id's broadcast) run on a process group of their own, created here by every rank: a
        # rank that is still parked in one of them when the watchdog gives up must not pair up with the barriers and the completion gather
        # that follow on the default group.
        import threading
        obox = {}
        setup_group = dist.new_group() if world > 1 else None
        oth = threading.Thread(target=lambda: obox.update(h=batch.open_rccl(ctx, dist, dev, rank, world, group=setup_group)), daemon=True)
        oth.start()
        oth.join(timeout=float(os.environ.get("ARTGPU_BENCH_RCCL_TIMEOUT", "180")))
        if "h" in obox:
            rccl_handle = obox["h"]
        else:
            print(f"[bench rank {rank}] the RCCL communicator for artgpu_batch_complete did not come up: completing through torch.distributed", file=sys.stderr, flush=True)
            use_rccl_capi, rccl_stuck = False, True

    guard = StepGuard(rank)
    for _ in range(args.warmup):
        guard.run(step)
    barrier()
    # per-launch duration of the dominant kernel, HIP events on the launch stream.  For AMaZE with border >= 4 the demosaic call launches
    # amaze_stream_kernel plus ~20 us of bookkeeping (a 2-KB memset / copy of the tile lists and the arena kernel over the tiles the
    # stream could not take: none at this frame size), no border_interpolate2 -- so the stage events recorded around the call bracket
    # that kernel (+0.5 %) and are read after the timed region; the other demosaicers use the library's own event pair, which the host
    # has to wait for once per step.
    lib_timing = xtrans or method != capi.BAYER_AMAZE or border < 4
    if lib_timing:
        ctx.enable_timing(True)
    kernel_ms = []
    evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    barrier()
    t0 = time.perf_counter()
    evs[0].record(stream)
    def timed_step(k):
        cur["ev"] = stage_ev[k]
        step()
        if lib_timing:
            kernel_ms.append(ctx.timings().demosaic_ms)
    for k in range(args.steps):
        guard.run(timed_step, k)
    cur["ev"] = None
    evs[1].record(stream)
    barrier()
    t1 = time.perf_counter()
    if lib_timing:
        ctx.enable_timing(False)
    elif not guard.status:
        kernel_ms = [ev[0].elapsed_time(ev[1]) for ev in stage_ev]
    # completion step: all-gather of the 64-byte per-rank records (the batch's only collective),
    # elapsed = MAX over ranks
    cs = batch.checksum64([int(d_out[1][H // 2, W // 2].item())]) if not guard.status else 0
    steps_done = guard.done_timed(args.warmup)
    hard_exit = False
    if use_rccl_capi:
        # under a launcher (any N): through the C ABI, over the RCCL communicator opened before the timed region (artgpu_batch_complete)
        # (in a watchdog thread: should the gather ever hang on some node, every rank times out the same way and the measurement is
        # completed through torch.distributed instead of being lost)
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(r=batch.complete_batch_rccl(ctx, rccl_handle, dist, dev, rank, world, steps_done, guard.status, cs, t1 - t0)), daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("ARTGPU_BENCH_RCCL_TIMEOUT", "180")))
        if "r" in box:
            records, elapsed, gather_via = box["r"]
            if rccl_handle is not None:
                rccl_handle.close()
        else:
            print(f"[bench rank {rank}] artgpu_batch_complete over RCCL did not finish: completing through torch.distributed", file=sys.stderr, flush=True)
            records, elapsed = batch.complete_batch(dist if world > 1 else None, dev, rank, steps_done, guard.status, cs, t1 - t0)
            gather_via = "torch.distributed (rccl-capi timed out)"
            hard_exit = True
    else:
        records, elapsed = batch.complete_batch(dist if world > 1 else None, dev, rank, steps_done, guard.status, cs, t1 - t0)
        gather_via = ("torch.distributed" if world > 1 else "single process") + (" (rccl-capi set-up timed out)" if rccl_stuck else "")
        hard_exit = rccl_stuck

    mp = W * H / 1e6
    failed = [r["rank"] for r in records if r["status"] != 0]
    if failed:      # (StepGuard: the gather completed on every rank; nobody reports a throughput for an incomplete batch)
        if rank == 0:
            print(json.dumps(failure_line(records, failed, elapsed, mp, args, world)), flush=True)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3)
    stage_ms = {nm: round(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for ev in stage_ev), 4) for i, nm in enumerate(stage_names)}
    value = world * args.lanes * args.steps * mp / elapsed
    kern_ms = statistics.mean(kernel_ms)
    achieved = (W * H * ALGO_BYTES_PER_PX / 1e9) / (kern_ms / 1e3)

    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process; the figure is the mean per
    # launch of the separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command, committed under
    # profiles/ (KB units; uncorrected -- the kernel's accesses are 4 B per lane, see profiles/r1/README.md)
    traffic, traffic_src, traffic_raw, issue = None, None, None, None
    kname = "xtrans_tiles_kernel" if xtrans else "amaze_stream_kernel" if method == capi.BAYER_AMAZE else "rcd_stream_kernel"
    pmc_file = {"amaze_stream_kernel": "amaze_stream_pmc_summary.json", "rcd_stream_kernel": "rcd_stream_pmc_summary.json",
                "xtrans_tiles_kernel": "xtrans_tiles_pmc_summary.json"}[kname]
    # the newest round's counter summary of the kernel (each carries the digest of the sources it was taken from: `traffic_stale` below)
    rounds = sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()), key=lambda d: -int(d[1:])) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    pmc_rel = next((os.path.join("profiles", rn, pmc_file) for rn in rounds if os.path.exists(os.path.join(ROOT, "profiles", rn, pmc_file))),
                   os.path.join("profiles", "r6", pmc_file))
    pmc_path = os.path.join(ROOT, pmc_rel)
    traffic_stale = None
    full_size = (W, H) == ((11648, 8736) if xtrans else (W45, H45))
    if full_size and os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        # FETCH_SIZE on gfx950 reports half the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section): doubled here.
        # (Sanity check for AMaZE: the raw figure, 156 MB, is less than the 179 MB CFA plane that has to be read at least once.)
        traffic_raw = {"FETCH_SIZE_KB": pmc["FETCH_SIZE"]["mean_per_launch"], "WRITE_SIZE_KB": pmc["WRITE_SIZE"]["mean_per_launch"]}
        traffic = int((2 * pmc["FETCH_SIZE"]["mean_per_launch"] + pmc["WRITE_SIZE"]["mean_per_launch"]) * 1024)
        traffic_src = pmc_rel + f" (rocprofv3 --pmc passes of `bench.py --workload {args.workload}`, kernel {kname}; FETCH_SIZE doubled)"
        # the counters describe the kernel as it was when they were taken: the summary carries the digest of its sources
        from art_amd import srchash
        traffic_stale = pmc.get("_source_sha256") != srchash.kernel_source_sha256(kname)
        # The streaming demosaicers are bound by instruction issue, not by HBM: instruction counts of the same PMC passes priced with
        # the per-SIMD issue times measured by scripts/ubench/issue_mix.hip (vector 1.19 ns, scalar 1.3 ns when mixed, LDS 1.06 ns;
        # the classes add up on this chip), over the 1024 SIMDs -- the time the kernel cannot go below without fewer instructions.
        if "SQ_INSTS_VALU" in pmc and "SQ_INSTS_SALU" in pmc and "SQ_INSTS_LDS" in pmc:
            iv, isa, il = pmc["SQ_INSTS_VALU"]["mean_per_launch"], pmc["SQ_INSTS_SALU"]["mean_per_launch"], pmc["SQ_INSTS_LDS"]["mean_per_launch"]
            model_ms = (iv * 1.19 + isa * 1.3 + il * 1.06) / 1024 / 1e6
            issue = {"valu": int(iv), "salu": int(isa), "lds": int(il), "ns_per_instr_per_simd": {"valu": 1.19, "salu": 1.3, "lds": 1.06},
                     "issue_time_ms": round(model_ms, 3), "frac_of_kernel_ms": None, "source": pmc_rel + " + scripts/ubench/issue_mix.hip"}

    # measured device-copy bandwidth in the same run (the practical HBM ceiling next to the 8 TB/s datasheet peak):
    # a 716 MB device-to-device copy, read + write bytes over its HIP-event time
    copy_gbs = None
    if rank == 0:
        nb = W * H * 4
        src_t = torch.empty(nb, dtype=torch.float32, device=dev).normal_()
        dst_t = torch.empty_like(src_t)
        ce = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(2):
            dst_t.copy_(src_t)
        ce[0].record()
        for _ in range(5):
            dst_t.copy_(src_t)
        ce[1].record()
        torch.cuda.synchronize(dev)
        copy_gbs = 5 * 2 * nb * 4 / 1e9 / (ce[0].elapsed_time(ce[1]) / 1e3)
        del src_t, dst_t

    if issue is not None:
        issue["frac_of_kernel_ms"] = round(issue["issue_time_ms"] / kern_ms, 3)
    result = {
        "metric": ("megapixels/sec end-to-end (X-Trans+FTblockDN+tone), 100 MP X-Trans" if xtrans else
                   "megapixels/sec end-to-end (AMaZE+FTblockDN+tone), 45 MP Bayer"),
        "value": round(value, 2),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ((("X-Trans 3-pass Markesteijn (CIELab)" if xt_passes == 3 else "X-Trans 1-pass Markesteijn (YPbPr)") if xtrans else "AMaZE") + f" + getImage/matrix + ImProcFunctions::denoise (chroma noise-curve map, FTblockDN wavelet shrinkage luma 40 / "
                         f"chroma 15 / gamma 1.7 + DCT detail recovery 50"
                         + (", guided chroma smoothing r=3, NL-means 50/80" if smoothing else "")
                         + f") + exposure 0.3 EV + tone curve {args.tone.upper()}, {W}x{H} " + ("X-Trans" if xtrans else "Bayer RGGB") + f" fp32, {args.lanes} frame{'s' if args.lanes > 1 else ''} per GPU per step "
                         + ("(BASELINE configs[4])" if xtrans else "(per-frame pipe of BASELINE configs[3])" if smoothing else "(BASELINE configs[2])"))
                        if pipeline else f"{args.workload.upper()} demosaic only, {W}x{H} Bayer RGGB fp32 (BASELINE configs[1])",
            "stage_ms": stage_ms,
            "stage_ms_schema": 2 if (pipeline and fused_tool) else 1,     # 1: one key per reference stage; 2: the fused tool's three stages as one key
            "stage_calls": ("fused tool: artgpu_improc_denoise_fused (getImage + matrix in front of ImProcFunctions::denoise, exposure behind it, inside its pixel passes)"
                            if pipeline and fused_tool else "one call per stage"),
            "frame": f"{W}x{H}", "frames_per_step": world * args.lanes, "lanes_per_gpu": args.lanes,
            "parallelism": f"frame-per-gpu x{world}" + (f", {args.lanes} frames in flight per GPU" if args.lanes > 1 else ""),
            "completion_records": len(records), "completion_via": gather_via,
        },
        "roofline": {
            "bound": "hbm", "kernel": kname,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale, "traffic_raw_counters": traffic_raw,
            # the whole step against the same contract (MP/s x 16 B / 8 TB/s), next to the dominant kernel's fraction
            "end_to_end_frac": round(value * 1e6 * ALGO_BYTES_PER_PX / 1e9 / HBM_PEAK_GBS / max(world, 1), 5),
            # the survey's diagnostic ceiling: traffic no fusion can remove because of global dependencies -- MAD medians need whole
            # subbands before shrinking, separable recurrences a transpose point: FTblockDN ~760 B/px on top of the 16 B/px contract (config 4: + NL-means
            # 12 B/px + guided filter ~40 B/px) => ~10 300 MP/s per GPU for config 3.  Not the contract figure; it says how far the chain is from what
            # its barriers allow.
            "barrier_aware": {"bytes_per_px": BARRIER_AWARE_B_PER_PX[args.workload], "ceiling_mps": round(HBM_PEAK_GBS * 1e9 / BARRIER_AWARE_B_PER_PX[args.workload] / 1e6, 1),
                              "end_to_end_frac": round(value / max(world, 1) / (HBM_PEAK_GBS * 1e9 / BARRIER_AWARE_B_PER_PX[args.workload] / 1e6), 4)},
            "kernel_ms": round(kern_ms, 4), "algorithmic_bytes": W * H * ALGO_BYTES_PER_PX,
            "device_copy_gbs": None if copy_gbs is None else round(copy_gbs, 1),
            "traffic_rate_frac_of_copy": None if (copy_gbs is None or traffic is None) else round(traffic / 1e9 / (kern_ms / 1e3) / copy_gbs, 4),
            "instruction_issue": issue,
        },
    }

    # the same step back to back for several seconds (no per-step barrier: the host runs ahead of the device, as a batch would)
    if args.sustained_seconds > 0:
        barrier()
        s0 = time.perf_counter()
        ns = 0
        chunk = max(args.steps, 10)
        while True:
            for _ in range(chunk):
                step()
            torch.cuda.synchronize(dev)
            ns += chunk
            if time.perf_counter() - s0 >= args.sustained_seconds:
                break
        s_el = time.perf_counter() - s0
        s_ms = 1e3 * s_el / ns
        result["sustained"] = {"seconds": round(s_el, 2), "steps": ns, "ms_per_step": round(s_ms, 4),
                               "value": round(args.lanes * mp / (s_ms / 1e3), 2), "unit": "MP/s per GPU",
                               "ratio_to_timed_steps": round(s_ms / (1e3 * elapsed / args.steps), 4)}

    # BASELINE.md C5 names both Markesteijn methods: the timed steps run THREE_PASS, the same step with ONE_PASS is reported beside it
    if xtrans and pipeline and world == 1 and args.lanes == 1 and xt_passes == 3 and not args.no_extra_legs:
        xt_passes = 1
        step0()
        barrier()
        o0 = time.perf_counter()
        no_ = max(1, min(args.steps, 5))
        for _ in range(no_):
            step0()
        barrier()
        o_ms = 1e3 * (time.perf_counter() - o0) / no_
        xt_passes = 3
        result["xtrans_one_pass"] = {"ms_per_step": round(o_ms, 4), "value": round(mp / (o_ms / 1e3), 2), "unit": "MP/s", "steps": no_,
                                     "note": "the same step with the ONE_PASS method (xtrans_demosaic.cc:688-741; flag: --xtrans-passes 1)"}

    # ART's default tone-curve mode is NEUTRAL (curves.cc:854-1038), the headline line uses STD: report the NEUTRAL step beside it
    if pipeline and world == 1 and args.tone == "std" and args.lanes == 1 and not args.no_extra_legs:
        def step_neutral():
            ctx.demosaic_xtrans(xt_passes, xt_passes == 3, p_raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, out) if xtrans else ctx.demosaic_bayer(method, p_raw, filt, 1.0, border, out)
            if fused_tool:
                ctx.improc_denoise_fused(img, dn, ws, demosaiced=out, sx1=border, sy1=border, mul=mul, do_clip=True, cam_to_work=mat,
                                         exposure=(exp_scale, 0.0), ecomp=expcomp, calclum_mat=mat, noise_c_curve=ccurve)
            else:
                ctx.get_image(out, border, border, mul, True, mat, img)
                ctx.improc_denoise(img, dn, ws, ecomp=expcomp, calclum_mat=mat, noise_c_curve=ccurve)
                ctx.exposure(img, exp_scale, 0.0)
            ctx.tone_curve_neutral(img, lut, 1.0, ws, iws_n)
        step_neutral()
        barrier()
        n0 = time.perf_counter()
        nn = max(1, min(args.steps, 5))
        for _ in range(nn):
            step_neutral()
        barrier()
        n_ms = 1e3 * (time.perf_counter() - n0) / nn
        result["neutral_tone"] = {"ms_per_step": round(n_ms, 4), "value": round(mp / (n_ms / 1e3), 2), "unit": "MP/s", "steps": nn}

    # Round-over-round comparability: up to round 3 `stage_ms` had one key per reference stage (demosaic / get_image+matrix / denoise /
    # exposure / tone_curve); since round 4 the default step calls the fused tool and the three middle stages are one key.  The line carries
    # BOTH: `config.stage_ms` describes the timed steps (schema 2 when the fused tool runs), `separate_stages` times a few steps with one
    # call per stage and reports the round-1..3 keys -- the difference between the two ms_per_step figures is what the API-level fusion is
    # worth, as opposed to kernel work.
    if pipeline and fused_tool and world == 1 and args.tone == "std" and args.lanes == 1 and not args.no_extra_legs:
        sep_names = ["demosaic", "get_image+matrix", "denoise", "exposure", "tone_curve"]
        ns_ = max(1, min(args.steps, 5))
        sev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(sep_names) + 1)] for _ in range(ns_)]

        def step_separate(ev=None):
            def mk(k):
                if ev is not None:
                    ev[k].record(stream)
            mk(0)
            ctx.demosaic_xtrans(xt_passes, xt_passes == 3, p_raw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, out) if xtrans else ctx.demosaic_bayer(method, p_raw, filt, 1.0, border, out)
            mk(1)
            ctx.get_image(out, border, border, mul, True, mat, img)
            mk(2)
            ctx.improc_denoise(img, dn, ws, ecomp=expcomp, calclum_mat=mat, noise_c_curve=ccurve)
            mk(3)
            ctx.exposure(img, exp_scale, 0.0)
            mk(4)
            ctx.tone_curve(img, lut, 1.0, True)
            mk(5)
        step_separate()
        barrier()
        q0 = time.perf_counter()
        for k in range(ns_):
            step_separate(sev[k])
        barrier()
        q_ms = 1e3 * (time.perf_counter() - q0) / ns_
        result["separate_stages"] = {"ms_per_step": round(q_ms, 4), "value": round(mp / (q_ms / 1e3), 2), "unit": "MP/s", "steps": ns_,
                                     "stage_ms": {nm: round(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for ev in sev), 4) for i, nm in enumerate(sep_names)},
                                     "stage_calls": "one call per stage (the round-1..3 key set; flag: --separate-stages)"}

    # what `--lanes 2` gives (a second frame in flight on its own context / stream / host thread), beside the one-frame-per-step line
    if pipeline and world == 1 and args.lanes == 1 and args.sustained_seconds > 0:
        ln2 = make_lane(1)
        in_flight(ctx, True); in_flight(ln2["ctx"], True)

        def step2():
            t_ = threading.Thread(target=lane_frame, args=(ln2,))
            t_.start()
            step0()
            t_.join()
            if lane_errors:
                raise lane_errors[0]
        step2()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n2 = max(1, min(args.steps, 10))
        for _ in range(n2):
            step2()
        torch.cuda.synchronize(dev)
        ms2 = 1e3 * (time.perf_counter() - t0) / n2
        result["two_frames_in_flight"] = {"ms_per_step": round(ms2, 4), "frames_per_step": 2, "value": round(2 * mp / (ms2 / 1e3), 2), "unit": "MP/s",
                                          "steps": n2, "flag": "--lanes 2"}
        in_flight(ctx, False)
        del ln2

    # The boundary can also take HOST buffers.  The PCIe-inclusive rate of the same frame -- never `value` -- through artgpu_batch_run_io:
    # uint16 sensor data up (scaleColors on the device), 16-bit scanlines down (rgb2out's matrix path + getScanline), pinned memory, a frame's
    # copies on streams of their own beside its neighbours' kernels (scripts/pcie_batch.py has the other forms)
    if pipeline and not xtrans and world == 1 and args.lanes == 1 and args.tone == "std" and args.sustained_seconds > 0 and not smoothing:
        try:
            import ctypes as C_
            pp = capi.PipelineParams()
            pp.sensor = 0; pp.bayer_method = method; pp.filters = filt; pp.initial_gain = 1.0; pp.xtrans_passes = 3; pp.border = border
            pp.mul[:] = mul; pp.do_clip = 1; pp.has_cam_to_work = 1
            pp.cam_to_work[:] = [float(v) for v in mat.reshape(9)]
            pp.ws[:] = [float(v) for v in ws.reshape(9)]; pp.iws[:] = [float(v) for v in iws_n.reshape(9)]
            pp.denoise_enabled = 1; pp.denoise = dn
            pp.exposure_enabled = 1; pp.expcomp = expcomp; pp.black = 0.0
            pp.tone_enabled = 1; pp.tone_mode = 0
            pp.tone_lut = lut.ctypes.data_as(C_.POINTER(C_.c_float)); pp.white_point = 1.0
            pp.to_out[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]; pp.to_work[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
            pp.scale = 1.0
            nio = 12
            raw16 = torch.from_numpy(np.clip(raw, 0, 65535).astype(np.uint16).view(np.int16)).pin_memory()
            scans = [torch.empty((ih, iw, 3), dtype=torch.int16).pin_memory() for _ in range(nio)]
            ins = [capi.sensor_frame(raw16.numpy().view(np.uint16)) for _ in range(nio)]
            outs = [capi.scanline_frame(t_.numpy().view(np.uint16), np.eye(3, dtype=np.float32)) for t_ in scans]
            legs = {}
            for nl in (1,):          # (one lane is the steady configuration: with two or three the figure moves by +-15 % between boxes and runs)
                ctx.set_batch_lanes(nl)
                ctx.batch_run_io(ins[:2 * nl], pp, outs[:2 * nl])
                h0 = time.perf_counter()
                ctx.batch_run_io(ins, pp, outs)
                h_ms = 1e3 * (time.perf_counter() - h0) / nio
                legs[f"lanes{nl}"] = {"ms_per_frame": round(h_ms, 3), "value": round(mp / (h_ms / 1e3), 1)}
            ctx.set_batch_lanes(1)
            result["host_buffers"] = {"unit": "MP/s", "frames": nio, "bytes_per_px": {"up": 2, "down": 6}, "host_memory": "pinned", **legs,
                                      "note": "PCIe-inclusive, artgpu_batch_run_io (uint16 sensor data in, 16-bit scanlines out); not the reported value"}
            del raw16, scans, ins, outs
        except Exception as e:      # noqa: BLE001  (a leg beside the line, never a reason to lose the line)
            result["host_buffers"] = {"error": repr(e)[:200]}
            try:
                ctx.set_batch_lanes(1)
            except Exception:      # noqa: BLE001
                pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib
        ncores = os.cpu_count() or 1
        os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
        cw, ch = W, H
        if pipeline:
            # bounded sample (~10-30 s of CPU work): the top-left quarter of the frame (same scene statistics)
            cw, ch = ((W // 2) // 6 * 6, (H // 2) // 6 * 6) if xtrans else ((W // 2) & ~1, (H // 2) & ~1)
            if xtrans:
                cw, ch = cw // 2 // 6 * 6, ch // 2 // 6 * 6          # 1/16 of the 100 MP frame keeps the CPU leg bounded
            craw = np.ascontiguousarray(raw[:ch, :cw])
            ciw, cih = cw - 2 * border, ch - 2 * border

            def fn():
                pl = (oracle_lib.xtrans_demosaic(craw, synth.XTRANS_FUJI, synth.XTRANS_RGB_CAM, 3, True) if xtrans
                      else oracle_lib.amaze(craw, filt, 1.0, border))
                im = oracle_lib.get_image(pl, border, border, ciw, cih, mul, True)
                im = oracle_lib.convert_color_space(im, mat)
                im = oracle_lib.improc_denoise(im, calclum_mat=mat, noise_c_curve=ccurve, smoothing=smoothing, radius=3,
                                               nl_strength=50 if smoothing else 0, nl_detail=80, ecomp=expcomp, ws=ws, detail_recovery=True)
                im = oracle_lib.exposure(im, exp_scale, 0.0)
                return oracle_lib.tone_neutral(im, lut, 1.0) if args.tone == "neutral" else oracle_lib.tone_std(im, lut, 1.0, True)
        else:
            fn = (lambda: oracle_lib.amaze(raw, filt, 1.0, 4)) if method == capi.BAYER_AMAZE else (lambda: oracle_lib.rcd(raw, filt))
        # Two timings of the same sample.  "checker": the oracle exactly as the tests use it (double-accumulated direct DCT, one core
        # per wavelet band).  The headline `cpu_baseline` is its TIMING variant (oracle_fast, never used as a checker): the 64-point DCTs
        # as an fp32 O(n log n) recursion -- the reference calls FFTW there, FTblockDN.cc:1604-1614 --, all cores on one band at a time
        # in the shrink passes, sliced MadRgb histograms: closer to what the reference's own OpenMP code does on this host.
        import ctypes
        fast_flag = ctypes.c_int.in_dll(oracle_lib.lib(), "oracle_fast")

        def timed(fast, reps):
            fast_flag.value = fast
            try:
                fn()  # warm-up (page faults)
                ts = []
                for _ in range(reps):
                    c0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - c0)
            finally:
                fast_flag.value = 0
            return statistics.median(ts)
        t_fast = timed(1, args.cpu_repeats)
        t_chk = timed(0, 1)
        sample = f"{cw}x{ch} region of the frame through the same stages of the CPU oracle (oracle/*.c, OpenMP), median of {args.cpu_repeats}"
        quarter = {"value": round(cw * ch / 1e6 / t_fast, 2), "unit": "MP/s", "cores": ncores, "kind": "port", "variant": "port-fast (timing only)", "sample": sample,
                   "checker": {"value": round(cw * ch / 1e6 / t_chk, 2), "unit": "MP/s", "cores": ncores, "kind": "port", "variant": "the tests' checker",
                               "sample": sample.replace(f"median of {args.cpu_repeats}", "one run")}}
        if pipeline and not args.cpu_sample_only:
            # ... and the WHOLE frame once, same inputs and stage boundaries as the timed GPU steps: the sample above is a
            # quarter (1/16 for X-Trans) of the frame and its figure moves with the host's mood (5.96 -> 4.95 MP/s between rounds 4 and 5 with no
            # change to the port); ~9 s of host time for 45 MP, ~25 s for 100 MP
            cw, ch, craw = W, H, raw
            ciw, cih = W - 2 * border, H - 2 * border
            fast_flag.value = 1
            try:
                c0 = time.perf_counter(); fn(); t_full = time.perf_counter() - c0
            finally:
                fast_flag.value = 0
            result["cpu_baseline"] = {"value": round(W * H / 1e6 / t_full, 2), "unit": "MP/s", "cores": ncores, "kind": "port", "variant": "port-fast (timing only)",
                                      "sample": f"the whole {W}x{H} frame through the same stages of the CPU oracle (oracle/*.c, OpenMP), one run of {round(t_full, 1)} s",
                                      "region_sample": quarter}
        else:
            result["cpu_baseline"] = quarter
        result["cpu_baseline"]["note"] = ("a CPU restatement of the reference's algorithm, not the reference (it cannot be built here: glibmm, lcms2, fftw3 are absent); "
                                          "SURVEY probe for scale: the reference's AMaZE alone runs 38.9 MP/s on 8 vCPU.  The GPU/CPU ratio is not a quality measure; "
                                          "roofline.frac and roofline.end_to_end_frac are.")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if hard_exit:               # a thread is still blocked inside RCCL: do not wait for it
        sys.stdout.flush()
        os._exit(0)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
