#!/usr/bin/env python3
"""bench.py -- ORB detect+describe frames/s on N MI355X GPUs (one process per GPU).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], batched as configs[2]): KITTI-shaped
1241x376 8-bit frames, 8 pyramid levels, scale 1.2, 1000 features, FAST-9
threshold 20, 3x3 NMS, Harris top-N, orientation patch 31, rotated BRIEF-256,
5x5 Gaussian blur on every level.  A "step" = one pass of the whole path over
one batch of --batch (default 64) synthetic frames that are already resident
in HBM; results stay resident in HBM too (D2H-inclusive rate is reported
separately as `fps_with_d2h`, it is never `value`).

Frames are independent, so ranks shard the stream with NO data-path collective
(weak scaling: every rank processes its own batch); torch.distributed (RCCL) is
used only for the barrier, the max-over-ranks time and a result checksum.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per launch (KITTI workload, batch 64) from separate rocprofv3 --pmc passes,
# see profiles/r01/pmc_traffic.md for the raw counters and the correction applied
PMC_TRAFFIC = {"k_blur": 209.4e6, "k_fast_nms": 28.4e6, "k_fast_nms_full_work": 79.3e6}


def stream_a(n, first=0):
    """SURVEY.md §8(d) stream A: frames derived deterministically from the two
    reference KITTI fixtures (roll + small Gaussian noise), KITTI statistics kept."""
    import oracle_lib as O  # only for load_kitti (reads tests/golden/*.npz)

    base = [O.load_kitti(0), O.load_kitti(1)]
    out = np.empty((n,) + base[0].shape, np.uint8)
    for j in range(n):
        i = first + j
        rng = np.random.default_rng(1000 + i)
        img = np.roll(base[i & 1], ((3 * i) % 17, (5 * i) % 11), (0, 1)).astype(np.int16)
        img += np.rint(rng.normal(0.0, 2.0, img.shape)).astype(np.int16)
        out[j] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def stream_b(n, h, w, first=0):
    """SURVEY.md §8(d) stream B: synthetic frames of any resolution."""
    out = np.empty((n, h, w), np.uint8)
    for j in range(n):
        rng = np.random.default_rng(first + j)
        img = 89.0 + 30.0 * rng.standard_normal((h, w))
        for _ in range(400):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            ww, hh = int(rng.integers(4, 120)), int(rng.integers(4, 120))
            img[y0:y0 + hh, x0:x0 + ww] += rng.uniform(20, 120) * rng.choice([-1, 1])
        out[j] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out


def cpu_baseline(frames, params_kw, budget_s=12.0, max_frames=4096):
    """The CPU oracle (a port of orb_cpu.cpp + the orb.cpp orchestrator intent)
    timed single-threaded on a bounded sample of the same workload."""
    import oracle_lib as O

    op = O.gpu_params(**params_kw)
    O.detect_and_compute_gpu(frames[0], op)  # warm
    t0 = time.perf_counter()
    done = 0
    while done < max_frames:  # cycles through the batch until the time budget is used
        O.detect_and_compute_gpu(frames[done % len(frames)], op)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d stream-A frames 1241x376 (the step's batch, cycled), same parameters, "
                     "oracle/liborb_oracle.so single thread, %.1f s" % (done, dt)}
    # the same oracle frame-parallel over the host cores this process may use (the
    # reference itself is single-threaded, so this is the most a frame-parallel CPU run of it could
    # give); ctypes releases the GIL during the call
    import threading

    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    nthr = max(1, min(ncores, 32))
    counts = [0] * nthr
    stop = time.perf_counter() + budget_s / 2

    def work(t):
        i = t
        while time.perf_counter() < stop:
            O.detect_and_compute_gpu(frames[i % len(frames)], op)
            counts[t] += 1
            i += nthr

    t1 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt2 = time.perf_counter() - t1
    out["all_cores"] = {"value": sum(counts) / dt2, "unit": "frames/s", "cores": nthr,
                        "sample": "%d frames over %d threads, %.1f s" % (sum(counts), nthr, dt2)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per step per GPU")
    ap.add_argument("--workload", default="kitti", choices=["kitti", "1080p"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true",
                    help="do not record per-stage HIP events inside the timed region (diagnostic)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl == RCCL; gloo only to rehearse N>1 on a one-GPU box)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--only-timed", action="store_true",
                    help="skip every extra pass (full-work FAST, D2H, single-frame, matcher): for rocprofv3 runs whose "
                         "per-kernel averages must describe the timed configuration only")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    pkg = importlib.import_module("visual-odometry-gpu_amd")
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    B = args.batch
    if args.workload == "kitti":
        H, W = 376, 1241
        pk = dict(nfeatures=1000, nlevels=8, scale_factor=1.2, threshold=20, n=9, nms_window=3, patch_size=31,
                  blur_levels=2, blur_kind=0)
        frames = stream_a(B, first=pkg.shard.frame_range(rank, world, B)[0])
        wl = "KITTI-shaped 1241x376 u8, 8 levels s=1.2, 1000 features, FAST-9 t=20, NMS 3x3, Harris top-N, blur5 all levels, BRIEF-256; batch=%d frames/step/GPU resident in HBM (stream A)" % B
    else:
        H, W = 1080, 1920
        pk = dict(nfeatures=4000, nlevels=12, scale_factor=1.2, threshold=20, n=9, nms_window=3, patch_size=31,
                  blur_levels=2, blur_kind=0)
        frames = stream_b(B, H, W, first=pkg.shard.frame_range(rank, world, B)[0])
        wl = "1920x1080 u8, 12 levels, 4000 features, Harris+NMS; batch=%d (stream B)" % B

    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, device=local_rank, **pk)
    ctx = pkg.Context(p)
    plan = ctx.plan(W, H)
    pyr_px = int((plan["level_w"].astype(np.int64) * plan["level_h"]).sum())
    d_frames = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()

    grp = pkg.shard.Group(world, device=torch.device("cuda", local_rank) if args.dist_backend == "nccl" else None)

    def barrier():
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    def step():
        ctx.batch_device(d_frames.data_ptr(), B, W, H)
        ctx.wait()

    for _ in range(args.warmup):
        step()

    # timed region: exactly K steps.  HIP events (on the context's stream) bracket
    # only the two roofline kernels, blur and fast+nms, so that the event records
    # do not inflate `value` (7 % with events around all seven stages).
    ctx.enable_stage_timing(0 if args.no_stage_events else 2)
    roof_ms = {"blur": 0.0, "fast_nms": 0.0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        # steps are enqueued back to back on the context's stream (each has its own
        # event set); one wait at the end, then the barrier + synchronize
        ctx.batch_device(d_frames.data_ptr(), B, W, H)
    ctx.wait()
    barrier()
    dt = time.perf_counter() - t0
    dt = grp.max_float(dt)
    fast_tiles = ctx.fast_tile_counts()  # (did the full work, all) of the last timed step
    if not args.no_stage_events:
        nread = min(args.steps, 64)
        for back in range(nread):
            lt = ctx.last_stage_times(back)
            roof_ms["blur"] += lt["blur"] / nread
            roof_ms["fast_nms"] += lt["fast_nms"] / nread
    # full per-stage breakdown from a separate, untimed pass
    nb = max(1, min(args.steps, 10))

    def breakdown():
        ctx.enable_stage_timing(1)
        acc = {k: 0.0 for k in pkg.orbx.STAGE_NAMES}
        for _ in range(nb):
            step()
            for k, v in ctx.last_stage_times().items():
                acc[k] += v / nb
        ctx.enable_stage_timing(0)
        return acc

    stage_ms = breakdown() if not args.only_timed else {k: 0.0 for k in pkg.orbx.STAGE_NAMES}
    full_ms = dict(stage_ms)
    if not args.only_timed:
        # the FAST kernel with its early exit switched off: every tile does the full work
        ctx.set_fast_early_exit(False)
        step()
        full_ms = breakdown()
        ctx.set_fast_early_exit(True)
    if args.no_stage_events:
        roof_ms = {k: stage_ms[k] for k in roof_ms}

    # D2H-inclusive rate (reported beside, never as `value`)
    cap = plan["out_capacity"]
    nd2h = 1 if args.only_timed else max(1, min(args.steps, 5))
    t1 = time.perf_counter()
    for _ in range(nd2h):
        ctx.batch_device(d_frames.data_ptr(), B, W, H)
        res = ctx.batch_fetch(0, B, cap)
    dt_d2h = (time.perf_counter() - t1) / nd2h

    # informational (never `value`): two contexts, each with its own stream and pools, fed alternately --
    # the latency-bound tail of one step (selection, describe) overlaps the front of the next one
    two_ctx_fps = None
    if rank == 0 and not args.only_timed:
        with pkg.Context(p) as ctx2:
            pair = (ctx, ctx2)
            for i in range(4):
                pair[i & 1].batch_device(d_frames.data_ptr(), B, W, H)
            ctx.wait()
            ctx2.wait()
            nst = max(args.steps, 10)
            t5 = time.perf_counter()
            for i in range(nst):
                pair[i & 1].batch_device(d_frames.data_ptr(), B, W, H)
            ctx.wait()
            ctx2.wait()
            two_ctx_fps = B * nst / (time.perf_counter() - t5)

    # next row, informational: Hamming 2-NN + ratio test of every consecutive
    # frame pair of the batch, on the device-resident descriptors (not part of `value`)
    match_ms, n_matches = 0.0, 0
    if not args.only_timed and B > 1:
        ctx.batch_device(d_frames.data_ptr(), B, W, H)
        ctx.wait()
        t3 = time.perf_counter()
        for _ in range(5):
            ctx.batch_match_consecutive(0.8)
        ctx.wait()
        match_ms = (time.perf_counter() - t3) / 5 * 1e3
        n_matches = len(ctx.batch_match_fetch(0, cap)[0])

    # Lucas-Kanade tracking between consecutive frames, the reference's call shape
    # (calcOpticalFlowPyrLK 21x21 / 3 levels / 30 it. / 0.01, feature_tracking.cpp:175-181): host frame in,
    # tracked points out, synchronous; the previous frame's pyramid stays on the device (prev=None)
    lk = None
    if rank == 0 and not args.only_timed and B > 1:
        kps0, _ = ctx.fast(frames[0], 20, 9, 3, 3000)
        pts0 = kps0.astype(np.float32)
        ctx.lk_track(frames[0], frames[1 % B], pts0)
        nlk = 50
        t4 = time.perf_counter()
        tracked = 0
        for i in range(nlk):
            _, st_lk, _ = ctx.lk_track(None, frames[(i + 2) % B], pts0)
            tracked += int(st_lk.sum())
        lk_ms = (time.perf_counter() - t4) / nlk * 1e3
        lk = {"ms_per_frame": lk_ms, "points": int(len(pts0)), "tracked_mean": tracked / nlk,
              "what": "orbx_lk_track: next frame in (host), tracked points out, cached previous pyramid, synchronous"}

    # per-frame host-in / host-out call (orbx_detect_and_compute, the reference's own call shape:
    # H2D of the frame + the whole path + one D2H of the results + sync), BASELINE.json configs[1]
    single = None
    if rank == 0 and not args.only_timed:
        p1 = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=1, device=local_rank, **pk)
        with pkg.Context(p1) as c1:
            for i in range(5):
                c1.detect_and_compute(frames[i % B])
            n1 = 200 if args.workload == "kitti" else 50
            t2 = time.perf_counter()
            for i in range(n1):
                r1 = c1.detect_and_compute(frames[i % B])
            dt1 = (time.perf_counter() - t2) / n1
            single = {"ms_per_frame": dt1 * 1e3, "frames_per_s": 1.0 / dt1, "keypoints": int(r1["count"]),
                      "what": "orbx_detect_and_compute: host frame in, host keypoints/descriptors out, synchronous"}

    # result checksum: same answer on every run / rank layout (frames are rank-specific)
    n_kp = grp.sum_int(int(res["counts"].sum()))
    csum = grp.sum_checksum(pkg.shard.descriptor_checksum(res["counts"], res["desc"]))

    if rank == 0:
        fps = world * B * args.steps / dt
        # dominant kernel among the two roofline stages.  Algorithmic
        # bytes: blur 2 B/px, FAST 1 B/px over all pyramid pixels.
        alg = {"blur": 2.0 * pyr_px * B, "fast_nms": 1.0 * pyr_px * B}
        dom = max(("blur", "fast_nms"), key=lambda k: roof_ms[k])

        def gbs(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        achieved = gbs(alg[dom], roof_ms[dom])
        both = gbs(alg["blur"] + alg["fast_nms"], roof_ms["blur"] + roof_ms["fast_nms"])
        # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/r01/pmc_traffic.md):
        # (FETCH_SIZE * c + WRITE_SIZE) * 1024, c = 1.35 calibrated on this access pattern
        traffic = {"k_blur": PMC_TRAFFIC["k_blur"], "k_fast_nms": PMC_TRAFFIC["k_fast_nms"]}
        out = {
            "metric": "ORB detect+describe frames/sec (1241x376, 8 lvls)" if args.workload == "kitti"
                      else "ORB detect+describe frames/sec (1920x1080, 12 lvls)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl, "frames_per_step_per_gpu": B, "sharding": "frame-parallel, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "k_blur2" if dom == "blur" else "k_fast_nms2",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (args.pmc_traffic if args.pmc_traffic is not None else
                                     (traffic["k_blur" if dom == "blur" else "k_fast_nms"]
                                      if args.workload == "kitti" and B == 64 else None)),
                         "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": roof_ms[dom],
                         "note": "k_fast_nms2 runs with its provable early exit in the timed region; "
                                 "full_work below is the same kernel with every tile doing the full work",
                         "blur": {"achieved": gbs(alg["blur"], roof_ms["blur"]),
                                  "frac": gbs(alg["blur"], roof_ms["blur"]) / HBM_PEAK_GBS,
                                  "avg_launch_ms": roof_ms["blur"]},
                         "fast_nms": {"achieved": gbs(alg["fast_nms"], roof_ms["fast_nms"]),
                                      "frac": gbs(alg["fast_nms"], roof_ms["fast_nms"]) / HBM_PEAK_GBS,
                                      "avg_launch_ms": roof_ms["fast_nms"]},
                         "fast_nms_full_work": {"achieved": gbs(alg["fast_nms"], full_ms["fast_nms"]),
                                                "frac": gbs(alg["fast_nms"], full_ms["fast_nms"]) / HBM_PEAK_GBS,
                                                "avg_launch_ms": full_ms["fast_nms"]},
                         "blur_plus_fast": {"achieved": both, "frac": both / HBM_PEAK_GBS,
                                            "algorithmic_bytes_per_step": alg["blur"] + alg["fast_nms"]}},
            "roofline_kernels_ms": roof_ms,
            "fast_tiles": {"full_work": fast_tiles[0], "total": fast_tiles[1],
                           "early_exit_frac": 1.0 - fast_tiles[0] / max(fast_tiles[1], 1)},
            "stage_ms_per_step": stage_ms,
            "fps_with_d2h": world * B / dt_d2h,
            "fps_two_contexts_alternating": two_ctx_fps,
            "single_frame_host_to_host": single,
            "lk_track": lk,
            "match_consecutive": {"ms_per_batch": match_ms, "pairs": B - 1, "matches_pair0": n_matches,
                                  "what": "Hamming 2-NN + 0.8 ratio test, frame i -> i+1, device-resident"},
            "keypoints_per_step": n_kp, "desc_checksum": csum,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, pk) if args.workload == "kitti" else None
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
