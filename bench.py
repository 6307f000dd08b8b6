#!/usr/bin/env python3
"""bench.py -- ORB detect+describe frames/s on N MI355X GPUs (one process per GPU).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], batched as configs[2]): KITTI-shaped 1241x376 8-bit frames,
8 pyramid levels, scale 1.2, 1000 features, FAST-9 threshold 20, 3x3 NMS, Harris top-N, orientation
patch 31, rotated BRIEF-256, 5x5 Gaussian blur on every level.  A "step" = one pass of the whole
path over one batch of --batch (default 1024) synthetic frames that are already resident in HBM; the
steps rotate over --rotate (default 2) DISTINCT resident batches, so the pools of a step (~3.5 GB) and
its inputs lie far beyond the 256 MiB Infinity Cache: every kernel streams from and to HBM.  Results stay resident in HBM too
(the D2H-inclusive rate is reported beside it as `fps_with_d2h`, it is never `value`).  The timed region is the
production shape: pipelined batches (orbx_set_pipelined_batches: two batches in flight on two lanes of the context),
one hipGraph launch per batch; per-kernel figures come from passes that run one batch at a time.

Frames are independent, so ranks shard the stream with NO data-path collective (weak scaling: every
rank processes its own batches); torch.distributed (RCCL) is used only for the barrier, the
max-over-ranks time and a result checksum.

--stream-frames F (BASELINE.json configs[3]): a synthetic KITTI stream of 8*F frames is split over
the ranks in contiguous blocks (shard.split_stream), each rank keeps its block resident in HBM and
walks it once in --batch-frame batches (distinct inputs every step); the all-reduced checksum is the
same for every world size.

Only the `cpu_baseline` / `parity` legs touch oracle/ (the checker); the timed path is liborbx.so.
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0      # same guide: measured float4 copy
VALU_CLOCK_GHZ = 2.4       # same guide: max clock; one wave64 VALU instruction holds its SIMD for 4 clocks
N_SIMD = 1024              # 256 CUs x 4
PMC_FILE = os.path.join(ROOT, "profiles", "r03", "pmc_counters.json")


def host_threads():
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    return max(1, min(ncores, 32))


def oracle_results(frames, params_kw):
    """The oracle on every frame (frame-parallel over the host cores; ctypes releases the GIL)."""
    import concurrent.futures as cf

    import oracle_lib as O

    op = O.gpu_params(**params_kw)
    O.lib()
    with cf.ThreadPoolExecutor(host_threads()) as ex:
        return list(ex.map(lambda f: O.detect_and_compute_gpu(f, op), frames))


def compare_with_oracle(pkg, frames, params_kw, counts, kps, desc):
    """keypoint lists and descriptor checksum of `frames` through the oracle against (counts, kps, desc)."""
    ref = oracle_results(frames, params_kw)
    want_cs = pkg.shard.descriptor_checksum([len(r["kps"]) for r in ref], [r["desc"] for r in ref])
    got_cs = pkg.shard.descriptor_checksum(counts, desc)
    kp_ok = all(int(counts[i]) == len(r["kps"]) and np.array_equal(kps[i, :len(r["kps"])], r["kps"]) for i, r in enumerate(ref))
    return {"checksum_match": bool(want_cs == got_cs), "keypoints_match": bool(kp_ok), "frames_checked": len(ref),
            "oracle_keypoints": int(sum(len(r["kps"]) for r in ref))}


def read_sclk_mhz():
    """Current shader clock from sysfs (the line of pp_dpm_sclk marked with *), or None."""
    import glob

    best = None
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if "*" in line:
                    mhz = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
                    best = mhz if best is None else max(best, mhz)
        except (OSError, ValueError, IndexError):
            pass
    return best


def cpu_baseline(frames, params_kw, budget_s=10.0, max_frames=4096):
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
           "sample": "%d frames %dx%d (the step's batch, cycled), same parameters, "
                     "oracle/liborb_oracle.so single thread, %.1f s" % (done, frames.shape[2], frames.shape[1], dt)}
    # the same oracle frame-parallel over the host cores this process may use (the
    # reference itself is single-threaded, so this is the most a frame-parallel CPU run of it could give)
    import threading

    nthr = host_threads()
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
    # BASELINE.json configs[0] / BASELINE.md §3: orb_cpu.cpp detectAndCompute on 000000.png, CPU-flavour
    # defaults (threshold 50, patch 9, cap 3000, ONE level), single thread
    import importlib

    k0 = importlib.import_module("visual-odometry-gpu_amd").streams.load_kitti(0)
    O.detect_and_compute_cpu(k0)
    t2, n0 = time.perf_counter(), 0
    while time.perf_counter() - t2 < 2.0:
        kps0 = O.detect_and_compute_cpu(k0)[0]
        n0 += 1
    dt3 = time.perf_counter() - t2
    out["config0_orb_cpu_000000"] = {"value": n0 / dt3, "unit": "frames/s", "cores": 1, "ms_per_frame": dt3 / n0 * 1e3,
                                     "keypoints": int(len(kps0)),
                                     "sample": "%d runs of the oracle's ORBCPU::detectAndCompute port on 000000.png "
                                               "(1241x376, 1 level, t=50, cap 3000)" % n0}
    return out


def kernel_source_sha():
    """Hash of the kernel sources: PMC counters from profiles/ are used only for the build they describe."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "visual-odometry-gpu_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(path):
    """Per-kernel counters of separate rocprofv3 --pmc passes (tools/pmc_collect.sh -> pmc_counters.json)."""
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    if d.get("source_sha") != kernel_source_sha():
        return None  # counters of another build say nothing about this one
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU (resident in HBM)")
    ap.add_argument("--rotate", type=int, default=2, help="distinct resident input batches the steps rotate over")
    ap.add_argument("--workload", default="kitti", choices=["kitti", "1080p"])
    ap.add_argument("--stream-frames", type=int, default=0,
                    help="config 3 as the TIMED region: a stream of 8*F frames split over the ranks, each rank walks its block once")
    ap.add_argument("--strong-frames", type=int, default=None,
                    help="config 3 beside the weak-scaling line (`stream` block): 8*F frames split over the ranks; "
                         "default 1000 (0 with --only-timed); 0: skip")
    ap.add_argument("--sustain-s", type=float, default=1.0, help="length of the sustained run (`value_sustained`); 0: skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl == RCCL; gloo only to rehearse N>1 on a one-GPU box)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--only-timed", action="store_true",
                    help="skip every extra pass (full-work FAST, D2H, single-frame, matcher): for rocprofv3 runs whose "
                         "per-kernel averages must describe the timed configuration only")
    ap.add_argument("--full-work", action="store_true",
                    help="run the TIMED region with the FAST early exit off (for rocprofv3 runs of the full-work kernel)")
    ap.add_argument("--unfused", action="store_true",
                    help="run the TIMED region with separate pyramid and blur kernels (for rocprofv3 runs of each kernel)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="timed region without orbx_set_pipelined_batches (consecutive batches then run one after the other "
                         "on one stream); implied by --only-timed, whose rocprofv3 kernel durations must not overlap")
    ap.add_argument("--pmc-file", default=PMC_FILE,
                    help="per-kernel counters from separate rocprofv3 --pmc passes of THIS build (tools/pmc_collect.sh)")
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
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    pkg = importlib.import_module("visual-odometry-gpu_amd")
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    B = args.batch
    stream_mode = args.stream_frames > 0
    if args.workload == "kitti":
        H, W = 376, 1241
        pk = dict(nfeatures=1000, nlevels=8, scale_factor=1.2, threshold=20, n=9, nms_window=3, patch_size=31,
                  blur_levels=2, blur_kind=0)
        wl = ("KITTI-shaped 1241x376 u8, 8 levels s=1.2, 1000 features, FAST-9 t=20, NMS 3x3, Harris top-N, blur5 all "
              "levels, BRIEF-256; batch=%d frames/step/GPU resident in HBM (stream A)" % B)
    else:
        H, W = 1080, 1920
        pk = dict(nfeatures=4000, nlevels=12, scale_factor=1.2, threshold=20, n=9, nms_window=3, patch_size=31,
                  blur_levels=2, blur_kind=0)
        wl = "1920x1080 u8, 12 levels, 4000 features, Harris+NMS; batch=%d (stream B)" % B

    # ---- inputs, resident in HBM before anything is timed
    if stream_mode:
        total = 8 * args.stream_frames
        first, last = pkg.shard.split_stream(total, rank, world)
        d_all = pkg.streams.stream_a_device(torch, first, last - first, dev)
        nlocal = last - first
        nbatches = (nlocal + B - 1) // B
        batches = [(d_all[i * B:min((i + 1) * B, nlocal)], min(B, nlocal - i * B)) for i in range(nbatches)]
        frames0 = batches[0][0].cpu().numpy()
        steps = args.steps if args.steps is not None else nbatches
        wl += "; stream mode: %d frames total, rank block %d" % (total, nlocal)
    else:
        R = max(1, args.rotate)
        steps = args.steps if args.steps is not None else 20
        f0 = pkg.shard.frame_range(rank, world, R * B)[0]
        if args.workload == "kitti":
            host = [pkg.streams.stream_a(B, first=f0 + r * B) for r in range(R)]
        else:
            host = [pkg.streams.stream_b(B, H, W, first=f0 + r * B) for r in range(R)]
        batches = [(torch.from_numpy(h).to(dev), B) for h in host]
        frames0 = host[0]
    torch.cuda.synchronize()

    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, device=local_rank, **pk)
    ctx = pkg.Context(p)
    plan = ctx.plan(W, H)
    cap = plan["out_capacity"]
    pyr_px = int((plan["level_w"].astype(np.int64) * plan["level_h"]).sum())
    grp = pkg.shard.Group(world, device=dev if args.dist_backend == "nccl" else None)

    def barrier():
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    def submit(i):
        t, n = batches[i % len(batches)]
        ctx.batch_device(t.data_ptr(), n, W, H)
        return n

    if args.full_work:
        ctx.set_fast_early_exit(False)
    if args.unfused:
        ctx.set_fused_pyramid_blur(False)
    for i in range(args.warmup):
        submit(i)
        ctx.wait()

    # ---- timed region: exactly `steps` steps, enqueued back to back, one wait at the end.  Production shape:
    # pipelined batches (orbx_set_pipelined_batches: consecutive batches alternate between two lanes -- own stream,
    # own pools -- and overlap), one hipGraph launch per batch, no event records.  With --only-timed / --no-pipeline:
    # one stream, and HIP events on it bracket the two roofline stages (kernel durations then do not overlap, which
    # the rocprofv3 collection needs).
    pipelined = not (args.only_timed or args.no_pipeline)
    if pipelined:
        ctx.set_pipelined_batches(True)
        for i in range(2):  # (both lanes' graphs captured before the clock starts)
            submit(i)
        ctx.wait()
        ctx.enable_stage_timing(0)
    else:
        ctx.enable_stage_timing(2)
    barrier()
    t0 = time.perf_counter()
    nframes = 0
    for i in range(steps):
        nframes += submit(i)
    ctx.wait()
    barrier()
    dt = time.perf_counter() - t0
    dt = grp.max_float(dt)
    nframes_all = grp.sum_int(nframes)
    fast_tiles = ctx.fast_tile_counts()  # (did the full work, all) of the last timed step
    pyr_done = ctx.pyramid_pixel_counts()  # (pyramid pixels produced, all) of the last timed step
    roof_ms = {"blur": 0.0, "fast_nms": 0.0}
    # ---- what the timed region itself produced: the result blocks of its LAST TWO steps (one per lane when
    # pipelined), copied out before anything else runs; compared with the oracle at the end (`parity.timed_region`)
    timed_out = []
    if not stream_mode and steps >= 2:
        for prev in (False, True):
            hv = ctx.batch_host_view(previous=prev)
            timed_out.append({"batch": (steps - 1 - (1 if prev else 0)) % len(batches), "counts": hv["counts"].copy(),
                              "kps": hv["kps"].copy(), "desc": hv["desc"].copy()})

    # ---- the same timed region with every FAST tile working and every pyramid row produced (`value_full_work`:
    # what a stream that never fills its caps in the top rows gets), and a sustained run (`value_sustained`)
    def timed_run(nsteps, nb=None):
        nb = len(batches) if nb is None else nb
        barrier()
        t_ = time.perf_counter()
        nfr = 0
        for i in range(nsteps):
            t, n = batches[i % nb]
            ctx.batch_device(t.data_ptr(), n, W, H)
            nfr += n
        ctx.wait()
        barrier()
        return grp.sum_int(nfr) / grp.max_float(time.perf_counter() - t_)

    value_full_work = value_sustained = sustained_info = None
    if not args.only_timed and not stream_mode and not args.full_work:
        ctx.set_fast_early_exit(False)
        for i in range(2):
            submit(i)
        ctx.wait()
        value_full_work = timed_run(steps)
        ctx.set_fast_early_exit(True)
        for i in range(2):
            submit(i)
        ctx.wait()
    if not args.only_timed and not stream_mode and args.sustain_s > 0:
        # >= 8 distinct resident batches (the graph cache holds 8 batch shapes: (input, result block) pairs)
        nb8 = 8
        base_first = pkg.shard.frame_range(rank, world, len(batches) * B)[0] + world * len(batches) * B
        while len(batches) < nb8:  # (the extra batches are generated on the device: nothing compares them with the oracle)
            k = len(batches)
            if args.workload == "kitti":
                batches.append((pkg.streams.stream_a_device(torch, base_first + k * B, B, dev), B))
            else:
                batches.append((torch.from_numpy(pkg.streams.stream_b(B, H, W, base_first + k * B)).to(dev), B))
        for i in range(nb8):
            submit(i)
        ctx.wait()
        rate = timed_run(2 * nb8)  # calibration
        nst = max(2 * nb8, int(args.sustain_s * rate / (world * B)) // nb8 * nb8)
        clocks = []
        import threading

        stop = threading.Event()

        def sample():
            while not stop.is_set():
                c_ = read_sclk_mhz()
                if c_:
                    clocks.append(c_)
                stop.wait(0.1)

        th = threading.Thread(target=sample)
        th.start()
        value_sustained = timed_run(nst)
        stop.set()
        th.join()
        sustained_info = {"steps": nst, "distinct_resident_batches": nb8, "seconds": nst * B * world / value_sustained,
                          "sclk_mhz_sysfs_median": (sorted(clocks)[len(clocks) // 2] if clocks else None),
                          "sclk_samples": len(clocks)}
        del batches[max(1, args.rotate):]
        for i in range(2):
            submit(i)
        ctx.wait()
    if pipelined:
        ctx.set_pipelined_batches(False)  # the per-stage passes below run one batch at a time
    else:
        nread = min(steps, 64)
        for back in range(nread):
            lt = ctx.last_stage_times(back)
            for k in roof_ms:
                roof_ms[k] += lt[k] / nread

    # ---- stream mode: the checksum of the whole pass (independent of the sharding)
    stream_info = None
    if stream_mode:
        ctx.enable_stage_timing(0)
        kp_local, cs_local = 0, 0
        for i in range(len(batches)):
            n = submit(i)
            res = ctx.batch_fetch(0, n, cap)
            kp_local += int(res["counts"].sum())
            cs_local = (cs_local + pkg.shard.descriptor_checksum(res["counts"], res["desc"])) & 0x7FFFFFFFFFFFFFFF
        stream_info = {"total_frames": 8 * args.stream_frames, "frames_this_rank": int(sum(b[1] for b in batches)),
                       "batches_this_rank": len(batches), "keypoints": grp.sum_int(kp_local),
                       "desc_checksum": grp.sum_checksum(cs_local),
                       "frames_per_s_strong": None}
        if steps == len(batches):  # the timed region was exactly one pass over every rank's block
            stream_info["frames_per_s_strong"] = stream_info["total_frames"] / dt

    # ---- BASELINE.json configs[3] beside the weak-scaling line: a stream of 8 x F distinct frames split over the
    # ranks in contiguous blocks, every rank keeps its block resident and walks it ONCE (strong scaling: the total
    # is fixed); the all-reduced keypoint count and descriptor checksum are the same for every world size
    # (N = 1 reference values: profiles/r03/README.md)
    def strong_scaling_block(F):
        total = 8 * F
        first, last = pkg.shard.split_stream(total, rank, world)
        d_all = pkg.streams.stream_a_device(torch, first, last - first, dev)
        nlocal = last - first
        sb = [(d_all[i * B:min((i + 1) * B, nlocal)], min(B, nlocal - i * B)) for i in range((nlocal + B - 1) // B)]
        torch.cuda.synchronize()
        ctx.enable_stage_timing(0)
        ctx.set_pipelined_batches(True)
        for t, n in sb[:2]:
            ctx.batch_device(t.data_ptr(), n, W, H)
        ctx.wait()
        barrier()
        t_ = time.perf_counter()
        for t, n in sb:
            ctx.batch_device(t.data_ptr(), n, W, H)
        ctx.wait()
        barrier()
        dt_ = grp.max_float(time.perf_counter() - t_)
        ctx.set_pipelined_batches(False)
        kp_local, cs_local = 0, 0
        for t, n in sb:
            ctx.batch_device(t.data_ptr(), n, W, H)
            res = ctx.batch_fetch(0, n, cap)
            kp_local += int(res["counts"].sum())
            cs_local = (cs_local + pkg.shard.descriptor_checksum(res["counts"], res["desc"])) & 0x7FFFFFFFFFFFFFFF
        del d_all, sb
        return {"total_frames": total, "frames_this_rank": nlocal, "frames_per_s_strong": total / dt_, "seconds": dt_,
                "keypoints": grp.sum_int(kp_local), "desc_checksum": grp.sum_checksum(cs_local),
                "what": "BASELINE.json configs[3]: 8 x %d stream-A frames (device generator), contiguous block per rank, "
                        "walked once, pipelined batches of %d; strong scaling" % (F, B)}

    strong_frames = args.strong_frames if args.strong_frames is not None else (0 if args.only_timed else 1000)
    if not stream_mode and strong_frames > 0 and args.workload == "kitti":
        stream_info = strong_scaling_block(strong_frames)

    # ---- untimed passes: all-stage breakdown with the timed configuration, then with every FAST tile working
    nb = max(1, min(steps, 10))

    def breakdown():
        ctx.enable_stage_timing(1)
        acc = {k: 0.0 for k in pkg.orbx.STAGE_NAMES}
        for i in range(nb):
            submit(i)
            ctx.wait()
            for k, v in ctx.last_stage_times().items():
                acc[k] += v / nb
        ctx.enable_stage_timing(0)
        return acc

    zero = {k: 0.0 for k in pkg.orbx.STAGE_NAMES}
    stage_ms = breakdown() if not args.only_timed else dict(zero)
    if pipelined:  # (the timed region carried no events: the roofline stages' times come from the per-stage pass)
        roof_ms = {"blur": stage_ms["blur"], "fast_nms": stage_ms["fast_nms"]}
    full_ms, full_tiles = dict(stage_ms), fast_tiles
    if not args.only_timed and not args.full_work:
        ctx.set_fast_early_exit(False)
        submit(0)
        ctx.wait()
        full_ms = breakdown()
        full_tiles = ctx.fast_tile_counts()
        ctx.set_fast_early_exit(True)
    if args.full_work:
        full_ms = dict(stage_ms) if not args.only_timed else dict(zero, fast_nms=roof_ms["fast_nms"], blur=roof_ms["blur"])
    # the pyramid and blur kernels on their own (the timed region runs them fused into one kernel)
    unfused_ms = None
    if not args.only_timed and not args.unfused:
        ctx.set_fused_pyramid_blur(False)
        submit(0)
        ctx.wait()
        unfused_ms = breakdown()
        ctx.set_fused_pyramid_blur(True)
    ctx.enable_stage_timing(0)

    # ---- results of batch 0 (for the checksum and the oracle comparison)
    n0 = submit(0)
    res0 = ctx.batch_fetch(0, n0, cap)
    kp0 = int(res0["counts"].sum())
    cs0 = pkg.shard.descriptor_checksum(res0["counts"], res0["desc"])

    # ---- D2H-inclusive rate (reported beside `value`, never as it): the result block of batch i is copied on
    # the context's copy stream while batch i+1 runs (orbx_batch_prefetch / orbx_batch_fetch_previous)
    fps_d2h = fps_d2h_blocking = fps_d2h_ref = None
    if not args.only_timed:
        # production shape (pipelined lanes) + orbx_batch_prefetch_compact: counts, packed keypoints, orientations and
        # descriptors of every batch land in the pinned mirror while the next batch runs and are read there in place
        nd = max(steps, 10)
        if pipelined:
            ctx.set_pipelined_batches(True)
        # orbx_set_host_results: the describe kernel writes the compact record (40 B per slot) into the pinned mirror
        # itself, so nothing is copied afterwards and orbx_batch_prefetch_compact only marks the block.  Measured
        # (tools/d2h_lag.py): the host link takes ~17 GB/s of such writes next to the kernels of the other lane (a
        # lone copy: 32 GB/s) -- ~430 k frames/s whatever the batch size, 0.99 of `value` at 256 frames per step and
        # 0.92 at 1024.  Copying the record after the batch instead (orbx_batch_prefetch_compact without it, the
        # runtime's copy kernel on the copy stream) gives 0.93-0.95 when the copy is enqueued after the wait for
        # the previous batch's results, and between 0.7 and 0.99 -- depending on the process's other streams -- when
        # it is enqueued right behind the batch, where it sits behind a wait for the batch's end.
        for i in range(2):
            submit(i)
        ctx.wait()
        t1 = time.perf_counter()
        for i in range(nd):  # the same loop with the results left on the device, for the ratio (it is not `value`)
            submit(i)
        ctx.wait()
        fps_d2h_ref = world * nd * batches[0][1] / (time.perf_counter() - t1)
        ctx.set_host_results(True)
        submit(0)
        ctx.batch_prefetch(compact=True)
        t1 = time.perf_counter()
        done = kp_seen = 0
        for i in range(1, nd + 1):
            n = submit(i)
            ctx.batch_prefetch(compact=True)
            hv = ctx.batch_host_view(previous=1)  # batch i-1, zero-copy from the pinned mirror
            done += len(hv["counts"])
            kp_seen += int(hv["counts"].sum())
        ctx.wait()
        fps_d2h = world * done / (time.perf_counter() - t1)
        ctx.set_host_results(False)
        if pipelined:
            ctx.set_pipelined_batches(False)
        t1 = time.perf_counter()
        done = 0
        for i in range(5):
            n = submit(i)
            ctx.batch_fetch(0, n, cap)
            done += n
        fps_d2h_blocking = world * done / (time.perf_counter() - t1)

    # ---- informational extras (never `value`)
    two_ctx_fps = lk = single = None
    match_ms, n_matches = 0.0, 0
    if rank == 0 and not args.only_timed and not stream_mode:
        with pkg.Context(p) as ctx2:  # two contexts fed alternately: the tail of one step overlaps the next one's front
            pair = (ctx, ctx2)
            t_, n_ = batches[0]
            for i in range(4):
                pair[i & 1].batch_device(t_.data_ptr(), n_, W, H)
            ctx.wait()
            ctx2.wait()
            nst = max(steps, 10)
            t5 = time.perf_counter()
            for i in range(nst):
                pair[i & 1].batch_device(batches[i % len(batches)][0].data_ptr(), n_, W, H)
            ctx.wait()
            ctx2.wait()
            two_ctx_fps = n_ * nst / (time.perf_counter() - t5)
        if B > 1:
            # next row: Hamming 2-NN + ratio test of consecutive frame pairs, device-resident
            submit(0)
            ctx.wait()
            t3 = time.perf_counter()
            for _ in range(5):
                ctx.batch_match_consecutive(0.8)
            ctx.wait()
            match_ms = (time.perf_counter() - t3) / 5 * 1e3
            n_matches = len(ctx.batch_match_fetch(0, cap)[0])
            # Lucas-Kanade between consecutive frames, the reference's call shape (feature_tracking.cpp:175-181)
            kps0, _ = ctx.fast(frames0[0], 20, 9, 3, 3000)
            pts0 = kps0.astype(np.float32)
            ctx.lk_track(frames0[0], frames0[1 % B], pts0)
            nlk, tracked = 50, 0
            t4 = time.perf_counter()
            for i in range(nlk):
                _, st_lk, _ = ctx.lk_track(None, frames0[(i + 2) % B], pts0)
                tracked += int(st_lk.sum())
            lk = {"ms_per_frame": (time.perf_counter() - t4) / nlk * 1e3, "points": int(len(pts0)),
                  "tracked_mean": tracked / nlk,
                  "what": "orbx_lk_track: next frame in (host), tracked points out, cached previous pyramid, synchronous"}
        # per-frame host-in / host-out call (orbx_detect_and_compute, the reference's own call shape), configs[1]
        p1 = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=1, device=local_rank, **pk)
        with pkg.Context(p1) as c1:
            for i in range(5):
                c1.detect_and_compute(frames0[i % B])
            n1 = 200 if args.workload == "kitti" else 50
            t2 = time.perf_counter()
            for i in range(n1):
                r1 = c1.detect_and_compute(frames0[i % B])
            dt1 = (time.perf_counter() - t2) / n1
            single = {"ms_per_frame": dt1 * 1e3, "frames_per_s": 1.0 / dt1, "keypoints": int(r1["count"]),
                      "what": "orbx_detect_and_compute: host frame in, host keypoints/descriptors out, synchronous"}

    n_kp = grp.sum_int(kp0)
    csum = grp.sum_checksum(cs0)

    if rank == 0:
        fps = nframes_all / dt
        # ---- roofline.  Algorithmic bytes (SURVEY.md §8d / BASELINE.md §4): blur 2 B/px, FAST 1 B/px over all
        # pyramid pixels; pyramid = level 0 read + every level written; per launch = per frame x batch.
        alg = {"pyramid": (W * H + pyr_px) * B, "blur": 2.0 * pyr_px * B, "fast_nms": 1.0 * pyr_px * B}
        pmc = load_pmc(args.pmc_file) if args.workload == "kitti" else None
        if pmc and pmc.get("batch") != B:
            pmc = None  # counters per launch of another batch size

        def gbs(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        def entry(name, nbytes, ms, pmc_key=None, extra=None):
            e = {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": nbytes}
            if nbytes:
                a = gbs(nbytes, ms)
                e.update({"achieved_GBps": a, "frac_of_8000": a / HBM_PEAK_GBS, "frac_of_6290_measured_copy": a / HBM_COPY_GBS})
            c = (pmc or {}).get("kernels", {}).get(pmc_key or name)
            if c:
                e["valu_insts_per_launch"] = c.get("valu")
                if c.get("valu"):
                    e["valu_floor_ms"] = c["valu"] * 4 / N_SIMD / (VALU_CLOCK_GHZ * 1e9) * 1e3
                if c.get("sq_active_inst_valu") and c.get("sq_busy_cycles"):
                    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs, SQ_BUSY_CYCLES cycles summed over the
                    # 32 shader engines: the share of the kernel's cycles in which a SIMD's vector unit was busy
                    e["valu_busy_frac"] = c["sq_active_inst_valu"] * 4 / N_SIMD / (c["sq_busy_cycles"] / 32)
                e["hbm_traffic_bytes_per_launch"] = c.get("traffic_bytes")
            if extra:
                e.update(extra)
            return e

        fast_kernel = "k_fast3" if os.environ.get("ORBX_FAST_IMPL") == "3" else "k_fast4"  # (the library's default: the streaming kernel)
        work_frac = fast_tiles[0] / max(fast_tiles[1], 1)
        pyr_frac = pyr_done[0] / max(pyr_done[1], 1)
        fused = not args.unfused
        kern = {}
        if fused:
            # the timed region builds and blurs the pyramid in ONE kernel.  Charged with the stage figures of
            # SURVEY.md §8d (resize + blur); `bytes_moved` = what the fused kernel has to move at all (level 0
            # read once + every blurred level written once).  With the early exit on the pyramid is built top
            # rows first and rows nobody reads are not produced: the timed-region entry charges only the
            # produced pixels; the every-row figure comes from the pass with the early exit off.
            moved = (W * H + pyr_px) * B
            charged = (W * H + (pyr_px - W * H)) * B + alg["blur"]
            if full_ms.get("blur", 0) > 0 and not args.only_timed:
                kern["pyramid_blur_fused_every_row"] = entry(
                    "pyrblur", charged, full_ms["blur"], "k_pyrblur_every_row",
                    {"compulsory_bytes_of_the_fused_kernel": moved, "achieved_GBps_compulsory": gbs(moved, full_ms["blur"]),
                     "note": "early exit off: one launch, every pyramid row produced; charged with SURVEY.md §8d's "
                             "resize + blur figures; the un-blurred pyramid is never written"})
            kern["pyramid_blur_fused_timed_region"] = entry(
                "pyrblur", charged * pyr_frac, roof_ms["blur"], "k_pyrblur",
                {"compulsory_bytes_of_the_fused_kernel": moved * pyr_frac,
                 "achieved_GBps_compulsory": gbs(moved * pyr_frac, roof_ms["blur"]),
                 "pyramid_pixels_produced": pyr_done[0], "pyramid_pixels": pyr_done[1],
                 "note": "production: top rows first (two launches, their time summed); bytes charged = produced "
                         "pixels / all pixels x (resize + blur figures of SURVEY.md §8d)"})
        src_ms = unfused_ms if fused else stage_ms
        if src_ms:
            kern["pyramid_alone"] = entry("pyramid", alg["pyramid"], src_ms["pyramid"], "k_pyramid2")
            kern["blur_alone"] = entry("blur", alg["blur"], src_ms["blur"] if fused else roof_ms["blur"], "k_blur3")
        kern.update({
            "fast_nms_full_work": entry("fast_nms", alg["fast_nms"], full_ms["fast_nms"], fast_kernel + "_full_work",
                                        {"tiles_worked": full_tiles[0], "tiles": full_tiles[1]}),
            "fast_nms_timed_region": entry("fast_nms", alg["fast_nms"] * work_frac, roof_ms["fast_nms"], fast_kernel,
                                           {"tiles_worked": fast_tiles[0], "tiles": fast_tiles[1],
                                            "note": "early exit on (production): bytes charged = tiles that worked / all tiles x 1 B/px"}),
            "select": entry("select", 0, stage_ms["select"] + stage_ms["compact"] + stage_ms["harris"], "k_level_select"),
            "describe": entry("describe", 0, stage_ms["describe"], "k_describe2"),
        })
        blur_ms = (unfused_ms or stage_ms)["blur"] if fused else roof_ms["blur"]
        # headline: the kernel of the two roofline stages (blur, FAST) that is furthest below the roof: FAST with
        # EVERY tile working (its early exit moves no bytes for the tiles it skips, so only this figure is a
        # bandwidth fraction); blur is the stand-alone kernel
        fracs = {"fast_nms_full_work": gbs(alg["fast_nms"], full_ms["fast_nms"])}
        if blur_ms > 0:
            fracs["blur_alone"] = gbs(alg["blur"], blur_ms)
        dom = min(fracs, key=lambda k: fracs[k] if fracs[k] > 0 else 1e30)
        dom_ms = kern[dom]["avg_launch_ms"] if dom in kern else full_ms["fast_nms"]
        achieved = fracs[dom]
        both_full = gbs(alg["blur"] + alg["fast_nms"], blur_ms + full_ms["fast_nms"])
        both_timed = gbs(alg["blur"] + alg["fast_nms"] * work_frac, blur_ms + roof_ms["fast_nms"])
        out = {
            "metric": "ORB detect+describe frames/sec (1241x376, 8 lvls)" if args.workload == "kitti"
                      else "ORB detect+describe frames/sec (1920x1080, 12 lvls)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong" if stream_mode else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl, "frames_per_step_per_gpu": B, "distinct_resident_batches": len(batches),
                       "sharding": "frame-parallel, no data-path collective",
                       "fast_early_exit": not args.full_work, "pyramid_blur_fused": not args.unfused,
                       "pyramid_top_rows_first": bool(pyr_done[0] < pyr_done[1]), "pipelined_batches": pipelined},
            "roofline": {"bound": "hbm", "kernel": "k_blur3 (stand-alone)" if dom == "blur_alone" else fast_kernel + (" (every tile working)" if fast_kernel == "k_fast3" else " (every unit working)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": kern.get(dom, {}).get("hbm_traffic_bytes_per_launch"),
                         "algorithmic_bytes_per_launch": alg["blur"] if dom == "blur_alone" else alg["fast_nms"],
                         "avg_launch_ms": dom_ms,
                         "residency": "batch %d: pools %.0f MB, %s the 256 MiB Infinity Cache" % (
                             B, 2 * 1.05 * pyr_px * B / 1e6 + W * H * B / 1e6,
                             "inside" if 2 * 1.05 * pyr_px * B + W * H * B < 256 * 2 ** 20 else "beyond"),
                         "blur_plus_fast": {"full_work": {"achieved": both_full, "frac": both_full / HBM_PEAK_GBS},
                                            "timed_region_bytes_touched": {"achieved": both_timed, "frac": both_timed / HBM_PEAK_GBS}},
                         "kernels": kern,
                         "pmc_counters": "profiles/r03/pmc_counters.json (same kernel sources)" if pmc else None},
            "roofline_kernels_ms": roof_ms,
            "fast_tiles": {"full_work": fast_tiles[0], "total": fast_tiles[1], "early_exit_frac": 1.0 - work_frac},
            "pyramid_pixels": {"produced": pyr_done[0], "total": pyr_done[1], "produced_frac": pyr_frac},
            "stage_ms_per_step": stage_ms,
            "stage_ms_per_step_fast_full_work": full_ms,
            "stage_ms_per_step_unfused": unfused_ms,
            "value_full_work": value_full_work, "value_sustained": value_sustained, "sustained": sustained_info,
            "fps_with_d2h": fps_d2h, "fps_with_d2h_blocking_fetch": fps_d2h_blocking,
            "fps_same_loop_without_copies": fps_d2h_ref, "d2h_bytes_per_frame": 4 + 40 * cap, "d2h_GBps_at_fps_with_d2h": (fps_d2h or 0) * (4 + 40 * cap) / 1e9,
            "fps_two_contexts_alternating": two_ctx_fps,
            "single_frame_host_to_host": single,
            "lk_track": lk,
            "match_consecutive": {"ms_per_batch": match_ms, "pairs": B - 1, "matches_pair0": n_matches,
                                  "what": "Hamming 2-NN + 0.8 ratio test, frame i -> i+1, device-resident"},
            "keypoints_batch0": n_kp, "desc_checksum": csum,
        }
        if stream_info:
            out["stream"] = stream_info
        if not args.no_cpu_baseline:
            # the oracle on the very frames of batch 0: the bench checks itself (src/compare.cpp:39-62 is the
            # reference's own, commented-out, CPU-vs-GPU descriptor check).  1920x1080: the first 8 frames.
            ncheck = n0 if args.workload == "kitti" else min(n0, 8)
            par = compare_with_oracle(pkg, frames0[:ncheck], pk, res0["counts"][:ncheck], res0["kps"][:ncheck], res0["desc"][:ncheck])
            par["what"] = ("frames of batch 0 through oracle/liborb_oracle.so: keypoint lists equal, descriptor checksum "
                           "equal" + (" (rank 0's batch)" if world > 1 else ""))
            # ... and on what the TIMED REGION itself left in its result blocks (pipelined lanes, graph replay): the
            # last two timed steps, one per lane
            if timed_out:
                tr = []
                for to in timed_out:
                    fr = host[to["batch"]]
                    nchk = len(fr) if args.workload == "kitti" else min(len(fr), 8)
                    r_ = compare_with_oracle(pkg, fr[:nchk], pk, to["counts"][:nchk], to["kps"][:nchk], to["desc"][:nchk])
                    r_["resident_batch"] = to["batch"]
                    tr.append(r_)
                par["timed_region"] = {"steps_checked": len(tr), "pipelined": pipelined,
                                       "checksum_match": all(r_["checksum_match"] for r_ in tr),
                                       "keypoints_match": all(r_["keypoints_match"] for r_ in tr),
                                       "frames_checked": sum(r_["frames_checked"] for r_ in tr), "per_step": tr,
                                       "what": "result blocks of the last two timed steps (one per lane), fetched before "
                                               "any other pass, against the oracle on their input batches"}
            out["parity"] = par
            out["cpu_baseline"] = cpu_baseline(frames0, pk) if args.workload == "kitti" else cpu_baseline(frames0[:8], pk, budget_s=6.0)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
