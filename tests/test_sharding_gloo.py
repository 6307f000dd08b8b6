"""N>1 path on CPU: two gloo ranks shard a frame stream exactly as bench.py does
under RCCL (visual-odometry-gpu_amd/shard.py), each rank runs its frames through
the CPU oracle (standing in for the GPU worker), and the reduced keypoint count
and descriptor checksum must equal a single-process run over the whole stream.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import importlib, os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import oracle_lib as O
shard = importlib.import_module("visual-odometry-gpu_amd.shard")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
grp = shard.Group(world)
per_rank = 2
first, last = shard.frame_range(rank, world, per_rank)
base = O.load_kitti(0)[60:260, 200:520]
p = O.gpu_params(nfeatures=200, nlevels=3)
counts, descs = [], []
for i in range(first, last):
    img = np.ascontiguousarray(np.roll(base, (3 * i, 5 * i), (0, 1)))
    r = O.detect_and_compute_gpu(img, p)
    counts.append(len(r["kps"])); descs.append(r["desc"])
grp.barrier()
dt = grp.max_float(0.5 + rank)          # MAX over ranks
n = grp.sum_int(sum(counts))
cs = grp.sum_checksum(shard.descriptor_checksum(counts, descs))
if rank == 0:
    print(json.dumps({"n": n, "cs": cs, "dt": dt, "world": world}))
dist.barrier(); dist.destroy_process_group()
"""


def run_world(world, port):
    code = WORKER % {"root": ROOT}
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json

    return json.loads(outs[0][0].strip().splitlines()[-1])


def test_two_gloo_ranks_equal_single_process(pkg):
    import oracle_lib as O

    two = run_world(2, 29611)
    assert two["world"] == 2 and two["dt"] == 1.5  # max over ranks of (0.5, 1.5)
    # single-process reference over the same 4 global frames
    base = O.load_kitti(0)[60:260, 200:520]
    p = O.gpu_params(nfeatures=200, nlevels=3)
    counts, descs = [], []
    for i in range(4):
        r = O.detect_and_compute_gpu(np.ascontiguousarray(np.roll(base, (3 * i, 5 * i), (0, 1))), p)
        counts.append(len(r["kps"]))
        descs.append(r["desc"])
    assert two["n"] == sum(counts) and two["n"] > 100
    assert two["cs"] == pkg.shard.Group(1).sum_checksum(pkg.shard.descriptor_checksum(counts, descs))


def test_split_helpers(pkg):
    s = pkg.shard
    assert [s.frame_range(r, 4, 64) for r in range(4)] == [(0, 64), (64, 128), (128, 192), (192, 256)]
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            parts = [s.split_stream(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        s.frame_range(4, 4, 1)
    # checksum is invariant under frame order / distribution
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, (3, 10, 32), dtype=np.uint8)
    c = np.array([10, 4, 0])
    a = s.descriptor_checksum(c, d)
    b = (s.descriptor_checksum(c[:1], d[:1]) + s.descriptor_checksum(c[1:], d[1:])) & 0x7FFFFFFFFFFFFFFF
    assert a == b and a != 0


def _run_bench_stream(world, port, frames_per_eighth):
    """bench.py --stream-frames with `world` gloo ranks, every rank on GPU 0 (the GPU worker, not the oracle)."""
    import json

    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--stream-frames",
            str(frames_per_eighth), "--batch", "16", "--warmup", "1", "--no-cpu-baseline", "--only-timed",
            "--dist-backend", "gloo", "--all-ranks-on-device0"]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    return json.loads(outs[0][0].strip().splitlines()[-1])


@pytest.mark.gpu
def test_two_gloo_ranks_with_the_gpu_worker_equal_one_rank(pkg):
    """BASELINE.json configs[3] in small: a 96-frame stream walked by ONE rank and by TWO ranks (contiguous
    blocks, shard.split_stream; both ranks on GPU 0, gloo for the reductions) gives the same keypoint
    count and the same all-reduced descriptor checksum -- through liborbx.so, not the oracle."""
    one = _run_bench_stream(1, 29641, 12)
    two = _run_bench_stream(2, 29642, 12)
    assert one["stream"]["total_frames"] == two["stream"]["total_frames"] == 96
    assert two["n_gpus"] == 2 and two["stream"]["frames_this_rank"] == 48
    assert one["stream"]["keypoints"] == two["stream"]["keypoints"] > 50000
    assert one["stream"]["desc_checksum"] == two["stream"]["desc_checksum"]


def _run_bench_default(world, port, strong):
    """bench.py's DEFAULT mode (what the driver launches: weak-scaling timed region) with the configs[3] `stream` block
    beside it, `world` gloo ranks on GPU 0."""
    import json

    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--batch", "16", "--steps", "4",
            "--warmup", "1", "--no-cpu-baseline", "--only-timed", "--strong-frames", str(strong),
            "--dist-backend", "gloo", "--all-ranks-on-device0"]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    return json.loads(outs[0][0].strip().splitlines()[-1])


@pytest.mark.gpu
def test_default_bench_line_carries_the_strong_scaling_block(pkg):
    """The line a driver `--gpus N` run gets is the weak-scaling one (`scaling: weak`); beside it the `stream` block
    reports BASELINE.json configs[3] (strong scaling: a fixed stream split over the ranks) with a keypoint count and a
    checksum that do not depend on the world size."""
    one = _run_bench_default(1, 29651, 12)
    two = _run_bench_default(2, 29652, 12)
    assert one["scaling"] == two["scaling"] == "weak" and two["n_gpus"] == 2
    assert one["stream"]["total_frames"] == two["stream"]["total_frames"] == 96
    assert two["stream"]["frames_this_rank"] == 48 and two["stream"]["frames_per_s_strong"] > 0
    assert one["stream"]["keypoints"] == two["stream"]["keypoints"] > 50000
    assert one["stream"]["desc_checksum"] == two["stream"]["desc_checksum"]
