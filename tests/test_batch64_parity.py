"""BASELINE.json configs[2] at full size: the benchmark's exact configuration -- 64 stream-A frames
(1241x376, 8 levels, scale 1.2, 1000 features, FAST-9 t=20, NMS 3x3, Harris top-N, blur on every
level), FAST early exit on, through `orbx_detect_and_compute_batch_device` -- compared frame by
frame with the oracle (restatement of src/orb_cpu.cpp:23-258 + the src/orb.cpp:58-109 orchestrator),
on the captured-graph launch path AND the plain launch path.  This is the check the reference left
commented out in src/compare.cpp:39-62 (CPU vs GPU descriptors of the same image).

Batch-dependent machinery exercised only here at the headline size: grid = frames x tiles in
band-major order, per-frame early-exit statistics, the graph cache, one result block for 64 frames.
"""
import concurrent.futures as cf
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["3", "4"], ids=["fast-tiles", "fast-stream"])
def fast_impl(request, monkeypatch):
    """Every test of this module runs with both FAST kernels of the whole path: the LDS tile kernel (orbx_fast.hip,
    the stage operators') and the register-streaming one (orbx_fast4.hip, the whole path's default); ORBX_FAST_IMPL is read when a context is created."""
    monkeypatch.setenv("ORBX_FAST_IMPL", request.param)
    return request.param

PK = dict(nfeatures=1000, nlevels=8, scale_factor=1.2, threshold=20, n=9, nms_window=3, patch_size=31,
          blur_levels=2, blur_kind=0)
B, W, H = 64, 1241, 376


@pytest.fixture(scope="module")
def frames(pkg):
    return pkg.streams.stream_a(B)


@pytest.fixture(scope="module")
def oracle_results(frames):
    op = O.gpu_params(**PK)
    O.lib()
    with cf.ThreadPoolExecutor(8) as ex:  # ctypes releases the GIL
        return list(ex.map(lambda f: O.detect_and_compute_gpu(f, op), frames))


def compare(res, refs, cap):
    assert res["counts"].shape == (B,)
    for i, ref in enumerate(refs):
        n = int(res["counts"][i])
        assert n == len(ref["kps"]), (i, n, len(ref["kps"]))
        assert np.array_equal(res["kps"][i, :n], ref["kps"]), i
        assert np.array_equal(res["kps_level"][i, :n], ref["kps_level"]), i
        assert np.array_equal(res["levels"][i, :n], ref["levels"]), i
        # north_star: angles / Harris within 1e-4 (they are in fact bit-identical)
        assert np.allclose(res["angles"][i, :n], ref["angles"], atol=1e-4, rtol=0), i
        assert np.allclose(res["responses"][i, :n], ref["responses"], rtol=1e-4, atol=1e-2), i
        assert np.array_equal(res["desc"][i, :n] & ref["valid"], ref["desc"] & ref["valid"]), i
        assert np.array_equal(res["desc"][i, :n], ref["desc"]), i  # the D15 bits follow the same zero-extension rule


@pytest.mark.parametrize("path", ["graph", "plain"])
def test_bench_configuration_matches_oracle(pkg, frames, oracle_results, path):
    import torch

    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        d = torch.from_numpy(frames).cuda()
        torch.cuda.synchronize()
        c.set_fast_early_exit(True)
        if path == "plain":
            c.enable_stage_timing(1)  # event records between the kernels: run_batch takes the plain launch path
        for _ in range(2):  # the second pass replays the captured graph / reuses the pools
            c.batch_device(d.data_ptr(), B, W, H)
            res = c.batch_fetch(0, B, cap)
        compare(res, oracle_results, cap)
        worked, total = c.fast_tile_counts()
        assert 0 < worked < total  # the early exit was really on
        # the checksum bench.py prints and compares
        shard = pkg.shard
        want = shard.descriptor_checksum([len(r["kps"]) for r in oracle_results], [r["desc"] for r in oracle_results])
        assert shard.descriptor_checksum(res["counts"], res["desc"]) == want


def test_full_work_equals_early_exit_at_batch_64(pkg, frames, oracle_results):
    import torch

    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        d = torch.from_numpy(frames).cuda()
        torch.cuda.synchronize()
        c.set_fast_early_exit(False)
        c.batch_device(d.data_ptr(), B, W, H)
        res = c.batch_fetch(0, B, cap)
        compare(res, oracle_results, cap)
        worked, total = c.fast_tile_counts()
        assert worked == total


_SWITCH_PROBE = r"""
import importlib, sys
import torch
sys.path.insert(0, %r)
pkg = importlib.import_module("visual-odometry-gpu_amd")
PK = %r
B, W, H = 64, 1241, 376
frames = pkg.streams.stream_a(B)
p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, **PK)
with pkg.Context(p) as c:
    cap = c.plan(W, H)["out_capacity"]
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    c.batch_device(d.data_ptr(), B, W, H)
    res = c.batch_fetch(0, B, cap)
    print("CHECKSUM", int(res["counts"].sum()), pkg.shard.descriptor_checksum(res["counts"], res["desc"]))
"""


@pytest.mark.parametrize("env", [{"ORBX_PYR_GROUP": "0", "ORBX_FAST_CHUNK": "1", "ORBX_TOP_ROWS": "0"},
                                 {"ORBX_PYR_GROUP": "7", "ORBX_FAST_CHUNK": "5", "ORBX_TOP_ROWS": "1"},
                                 {"ORBX_PYR_GROUP": "256", "ORBX_FAST_CHUNK": "64", "ORBX_TOP_ROWS": "3"}])
def test_dispatch_shape_switches_do_not_change_results(pkg, oracle_results, env):
    """ORBX_PYR_GROUP (frames per dispatch group of the fused pyramid + blur kernel), ORBX_FAST_CHUNK
    (tiles per FAST workgroup) and ORBX_TOP_ROWS (tile rows of the first pass of the top-rows-first
    pipeline) are read once per process: a child process per setting, same checksum
    as the oracle's (tests/test_abi.py lists them as result-preserving switches)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _SWITCH_PROBE % (root, PK)], env=e, capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("CHECKSUM")][-1].split()
    shard = pkg.shard
    want = shard.descriptor_checksum([len(r["kps"]) for r in oracle_results], [r["desc"] for r in oracle_results])
    assert int(line[1]) == sum(len(r["kps"]) for r in oracle_results)
    assert int(line[2]) == want


def test_top_rows_first_pipeline_with_sparse_and_bottom_heavy_frames(pkg, frames):
    """The pyramid is built top rows first and the rest of a level only if its top tile rows did not reach
    `cap` survivors (orbx_api.cpp, enqueue_batch).  Mixed batch: dense stream-A frames (rest skipped),
    frames whose corners all lie in the BOTTOM half (everything has to be produced, keypoints far below the
    first-pass rows), near-empty frames, and frames with a dense top band on some levels only -- every
    frame compared with the oracle."""
    import torch

    rng = np.random.default_rng(77)
    mixed = frames.copy()
    for i in range(B):
        kind = i % 4
        if kind == 1:  # corners only below row 200
            img = np.full((H, W), 90, np.uint8)
            for _ in range(400):
                x, y = int(rng.integers(8, W - 40)), int(rng.integers(200, H - 30))
                img[y:y + int(rng.integers(4, 24)), x:x + int(rng.integers(4, 30))] = int(rng.integers(0, 256))
            mixed[i] = img
        elif kind == 2:  # a handful of corners anywhere
            img = np.full((H, W), 128, np.uint8)
            for _ in range(6):
                x, y = int(rng.integers(8, W - 40)), int(rng.integers(8, H - 30))
                img[y:y + 12, x:x + 17] = 255
            mixed[i] = img
        elif kind == 3:  # fine texture in the top 60 rows (upper levels never reach their cap there), blocks below
            img = np.full((H, W), 60, np.uint8)
            img[:60] = rng.integers(0, 256, (60, W), dtype=np.uint8)
            for _ in range(150):
                x, y = int(rng.integers(8, W - 40)), int(rng.integers(70, H - 30))
                img[y:y + int(rng.integers(6, 20)), x:x + int(rng.integers(6, 20))] = int(rng.integers(0, 256))
            mixed[i] = img
    op = O.gpu_params(**PK)
    with cf.ThreadPoolExecutor(8) as ex:
        refs = list(ex.map(lambda f: O.detect_and_compute_gpu(f, op), mixed))
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        c.set_top_rows_first(1)  # always (the adaptive default may decide that this batch does not pay)
        d = torch.from_numpy(mixed).cuda()
        torch.cuda.synchronize()
        # a first batch of other frames leaves ITS rows in the pool: a strip skipped now must not be read
        d0 = torch.from_numpy(np.ascontiguousarray(mixed[::-1])).cuda()
        c.batch_device(d0.data_ptr(), B, W, H)
        c.wait()
        for _ in range(2):
            c.batch_device(d.data_ptr(), B, W, H)
            res = c.batch_fetch(0, B, cap)
            compare(res, refs, cap)
        done, total = c.pyramid_pixel_counts()
        assert 0 < done < total  # some levels skipped their lower rows, others did not
        c.set_fast_early_exit(False)  # one pass, every row
        c.batch_device(d.data_ptr(), B, W, H)
        compare(c.batch_fetch(0, B, cap), refs, cap)
        done, total = c.pyramid_pixel_counts()
        assert done == total


def test_top_rows_first_adapts_to_the_stream(pkg, frames, oracle_results):
    """Adaptive mode (default): the second pass reports how many levels it could skip; a stream on which it
    skips (almost) nothing falls back to one pass after the first batches, a dense stream keeps two passes.
    Results equal the oracle's throughout."""
    import torch

    sparse = np.full((B, H, W), 128, np.uint8)
    rng = np.random.default_rng(5)
    for i in range(B):
        for _ in range(8):
            x, y = int(rng.integers(8, W - 40)), int(rng.integers(8, H - 30))
            sparse[i, y:y + 12, x:x + 17] = 255
    op = O.gpu_params(**PK)
    with cf.ThreadPoolExecutor(8) as ex:
        refs = list(ex.map(lambda f: O.detect_and_compute_gpu(f, op), sparse))
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        ds, dd = torch.from_numpy(sparse).cuda(), torch.from_numpy(frames).cuda()
        torch.cuda.synchronize()
        fracs = []
        for _ in range(4):
            c.batch_device(ds.data_ptr(), B, W, H)
            compare(c.batch_fetch(0, B, cap), refs, cap)  # (fetch waits: the report of this batch has arrived)
            done, total = c.pyramid_pixel_counts()
            fracs.append(done / total)
        assert fracs[-1] == 1.0  # one pass by now (two-pass batches of this stream produce everything as well)
        c.set_top_rows_first(2)  # reset the verdict: a dense stream keeps its two passes
        for _ in range(3):
            c.batch_device(dd.data_ptr(), B, W, H)
            compare(c.batch_fetch(0, B, cap), oracle_results, cap)
            done, total = c.pyramid_pixel_counts()
        assert done < 0.7 * total


def test_pipelined_batches_overlap_without_changing_results(pkg, frames, oracle_results):
    """orbx_set_pipelined_batches: consecutive device-resident batches alternate between two lanes (own stream, own
    pools) and overlap on the GPU.  A stream of batches with three different inputs, results read from the pinned
    mirror of the PREVIOUS batch while the next one runs (the bench's streaming-consumer loop), must equal what the
    same batches give one at a time; entry points outside the streaming set may be called in between."""
    import torch

    inputs = [frames, np.ascontiguousarray(frames[::-1]), np.ascontiguousarray(np.roll(frames, (5, 9), (1, 2)))]
    dev = [torch.from_numpy(a).cuda() for a in inputs]
    torch.cuda.synchronize()
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, **PK)
    shard = pkg.shard
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        want = []
        for d in dev:  # one at a time
            c.batch_device(d.data_ptr(), B, W, H)
            r = c.batch_fetch(0, B, cap)
            want.append((int(r["counts"].sum()), shard.descriptor_checksum(r["counts"], r["desc"])))
        c.set_pipelined_batches(True)
        order = [0, 1, 2, 0, 2, 1, 1, 0, 2, 2]
        got = []
        c.batch_device(dev[order[0]].data_ptr(), B, W, H)
        c.batch_prefetch()
        for k in order[1:]:
            c.batch_device(dev[k].data_ptr(), B, W, H)
            hv = c.batch_host_view(previous=True)  # the batch before, zero-copy from its pinned mirror
            got.append((int(hv["counts"].sum()), shard.descriptor_checksum(hv["counts"], hv["desc"])))
            c.batch_prefetch()
        r = c.batch_fetch(0, B, cap)  # the last one (waits for its lane)
        got.append((int(r["counts"].sum()), shard.descriptor_checksum(r["counts"], r["desc"])))
        assert got == [want[k] for k in order]
        # the first input against the oracle as well, through the pipelined path
        c.batch_device(dev[1].data_ptr(), B, W, H)
        c.batch_device(dev[0].data_ptr(), B, W, H)
        compare(c.batch_fetch(0, B, cap), oracle_results, cap)
        # entry points outside the streaming set wait for both lanes: a stage operator and the tile counts in between
        c.batch_device(dev[2].data_ptr(), B, W, H)
        c.batch_device(dev[1].data_ptr(), B, W, H)
        kps, total = c.fast(inputs[0][0], 20, 9, 3, 3000)
        assert total > 0
        worked, alltiles = c.fast_tile_counts()
        assert 0 < worked <= alltiles
        c.wait()
        c.set_pipelined_batches(False)
        c.batch_device(dev[0].data_ptr(), B, W, H)
        compare(c.batch_fetch(0, B, cap), oracle_results, cap)
