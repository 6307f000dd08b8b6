"""What the benchmark line rests on, against the oracle, at the benchmark's own shapes:

* the HEADLINE shape: 256 stream-A frames per batch, two resident batches, pipelined lanes, one hipGraph replay per
  batch -- the result blocks of the last two steps of such a run (one per lane), exactly what bench.py's
  `parity.timed_region` looks at;
* BASELINE.json configs[4] as a BATCH: 1920x1080 stream-B frames, 12 levels, 4000 features through
  orbx_detect_and_compute_batch_device (three-kernel selection, adaptive fall-back of the top-rows-first pipeline);
* batches on a caller's stream mixed with pipelined ones (the lane's pools and the result block follow
  their previous user's event);
* the compact result copy (orbx_batch_prefetch_compact).
This is the check the reference left commented out in src/compare.cpp:39-62, at scale."""
import concurrent.futures as cf

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

PK = dict(nfeatures=1000, nlevels=8, scale_factor=1.2, threshold=20, n=9, nms_window=3, patch_size=31,
          blur_levels=2, blur_kind=0)
W, H = 1241, 376


def oracle_on(frames, pk, threads=16):
    op = O.gpu_params(**pk)
    O.lib()
    with cf.ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL
        return list(ex.map(lambda f: O.detect_and_compute_gpu(f, op), frames))


def compare(res, refs, full=True):
    # (copies: a zero-copy view of the pinned mirror dangles once the context is closed, and pytest prints the locals
    # of a failed assertion after that)
    res = {k: (None if v is None else np.array(v)) for k, v in res.items()}
    for i, ref in enumerate(refs):
        n = int(res["counts"][i])
        assert n == len(ref["kps"]), (i, n, len(ref["kps"]))
        if res.get("kps") is not None:
            assert np.array_equal(res["kps"][i, :n], ref["kps"]), i
        if res.get("kps16") is not None:  # (the packed pairs of the host views; the only ones of a compact copy)
            assert np.array_equal(res["kps16"][i, :n].astype(np.int32), ref["kps"]), i
        assert res.get("kps") is not None or res.get("kps16") is not None
        assert np.allclose(res["angles"][i, :n], ref["angles"], atol=1e-4, rtol=0), i
        assert np.array_equal(res["desc"][i, :n], ref["desc"]), i
        if full:
            assert np.array_equal(res["kps_level"][i, :n], ref["kps_level"]), i
            assert np.array_equal(res["levels"][i, :n], ref["levels"]), i
            assert np.allclose(res["responses"][i, :n], ref["responses"], rtol=1e-4, atol=1e-2), i


@pytest.fixture(scope="module")
def batches256(pkg):
    return [pkg.streams.stream_a(256, first=0), pkg.streams.stream_a(256, first=256)]


@pytest.fixture(scope="module")
def oracle256(batches256):
    return [oracle_on(b, PK) for b in batches256]


@pytest.mark.parametrize("impl", ["3", "4"], ids=["fast-tiles", "fast-stream"])
def test_headline_shape_pipelined_graph_replay_matches_oracle(pkg, batches256, oracle256, impl, monkeypatch):
    """bench.py's timed region: 256 frames per batch, 2 resident batches, pipelined lanes, graph replay.  After 20 steps
    the two result blocks (read from the pinned mirrors, one per lane) hold the last two steps: both must equal the
    oracle on their input batch."""
    import torch

    monkeypatch.setenv("ORBX_FAST_IMPL", impl)
    dev = [torch.from_numpy(b).cuda() for b in batches256]
    torch.cuda.synchronize()
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=256, **PK)
    with pkg.Context(p) as c:
        c.set_pipelined_batches(True)
        steps = 20
        for i in range(steps):
            c.batch_device(dev[i % 2].data_ptr(), 256, W, H)
        c.wait()
        last = c.batch_host_view()
        prev = c.batch_host_view(previous=True)
        compare(last, oracle256[(steps - 1) % 2])
        compare(prev, oracle256[(steps - 2) % 2])
        # with every FAST tile working / every pyramid row produced (bench.py's value_full_work): the same results
        c.set_fast_early_exit(False)
        for i in range(4):
            c.batch_device(dev[i % 2].data_ptr(), 256, W, H)
        c.wait()
        compare(c.batch_host_view(), oracle256[1])
        compare(c.batch_host_view(previous=True), oracle256[0])


def test_caller_stream_batches_between_pipelined_ones(pkg, batches256, oracle256):
    """Pipelining on; batches on a caller's stream (never pipelined: lane 0's pools) interleaved with batches on the
    context's own stream (lanes), without any host synchronisation in between.  Every batch must still be right:
    a batch that comes to a lane's pools or to a result block on another stream waits for their previous user."""
    import torch

    n = 64
    a = torch.from_numpy(batches256[0][:n]).cuda()
    b = torch.from_numpy(batches256[1][:n]).cuda()
    torch.cuda.synchronize()
    refs = {id(a): oracle256[0][:n], id(b): oracle256[1][:n]}
    s1 = torch.cuda.Stream()
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=n, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        c.set_pipelined_batches(True)
        # the sequence of ADVICE r02: caller's stream while blk == 0, then the context's stream -- and the mirror case
        seq = [(a, None), (b, s1), (a, None), (b, None), (a, s1), (b, s1), (a, None), (b, s1), (a, None)]
        for k, (t, st) in enumerate(seq):
            c.batch_device(t.data_ptr(), n, W, H, stream=st.cuda_stream if st is not None else None)
            if k in (3, 6, 8):  # (a fetch waits for the batch concerned only)
                compare(c.batch_fetch(0, n, cap), refs[id(t)])
        # back to back without any fetch, then both blocks
        order = [(a, s1), (b, None), (a, None), (b, s1), (a, None), (b, None)]
        for t, st in order:
            c.batch_device(t.data_ptr(), n, W, H, stream=st.cuda_stream if st is not None else None)
        c.wait()
        s1.synchronize()
        compare(c.batch_host_view(), refs[id(order[-1][0])])
        compare(c.batch_host_view(previous=True), refs[id(order[-2][0])])


def test_compact_prefetch_equals_full_view(pkg, batches256, oracle256):
    """orbx_batch_prefetch_compact copies counts | packed keypoints | orientations | descriptors only (40 bytes per
    slot); what it delivers equals the full copy, the other sections read NULL, a fetch unpacks the keypoints and one
    that asks for the other sections gets them all the same."""
    import torch

    n = 32
    d = torch.from_numpy(batches256[0][:n]).cuda()
    torch.cuda.synchronize()
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=n, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        c.batch_device(d.data_ptr(), n, W, H)
        c.batch_prefetch(compact=True)
        hv = c.batch_host_view()
        assert hv["responses"] is None and hv["levels"] is None and hv["kps_level"] is None and hv["kps"] is None
        assert hv["kps16"].dtype == np.uint16
        compare(hv, oracle256[0][:n], full=False)
        full = c.batch_fetch(0, n, cap)  # asks for every section: the rest is copied now
        compare(full, oracle256[0][:n])
        assert np.array_equal(hv["counts"], full["counts"])
        for i in range(n):  # (only the first counts[i] slots of a frame are written)
            m = int(full["counts"][i])
            for k in ("angles", "desc"):
                assert np.array_equal(hv[k][i, :m], full[k][i, :m]), k
            assert np.array_equal(hv["kps16"][i, :m].astype(np.int32), full["kps"][i, :m])
        # streaming use: compact copy of batch i overlaps batch i + 1
        c.set_pipelined_batches(True)
        c.batch_device(d.data_ptr(), n, W, H)
        c.batch_prefetch(compact=True)
        c.batch_device(d.data_ptr(), n, W, H)
        compare(c.batch_host_view(previous=True), oracle256[0][:n], full=False)
        c.batch_prefetch()  # the whole block of the last batch
        compare(c.batch_host_view(), oracle256[0][:n])
        # the ring holds four blocks: views reach up to three batches back
        for _ in range(2):
            c.batch_device(d.data_ptr(), n, W, H)
        for back in (0, 1, 2, 3):  # (the block three batches back still holds the compact copy made above)
            compare(c.batch_host_view(previous=back), oracle256[0][:n], full=back < 3)


@pytest.mark.parametrize("n", [32, 5])
def test_host_results_written_by_the_describe_kernel(pkg, batches256, oracle256, n):
    """orbx_set_host_results: the compact record reaches the pinned mirror from the describe kernel itself.  The host
    views and fetches deliver what the copies delivered -- compact mark, whole block, blocking fetch, pipelined
    streaming with the ring of four blocks, graph replay and plain launches, both describe variants (n = 5: one
    keypoint per wave) -- and switching it off again restores the copies."""
    import torch

    d = [torch.from_numpy(batches256[k][:n]).cuda() for k in range(2)]
    torch.cuda.synchronize()
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=n, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        c.set_host_results(True)
        c.batch_device(d[0].data_ptr(), n, W, H)
        c.batch_prefetch(compact=True)  # (nothing to copy)
        hv = c.batch_host_view()
        assert hv["kps"] is None and hv["responses"] is None
        compare(hv, oracle256[0][:n], full=False)
        compare(c.batch_fetch(0, n, cap), oracle256[0][:n])  # every section: the others are copied now
        c.batch_device(d[1].data_ptr(), n, W, H)
        c.batch_prefetch()  # the whole block: the sections the kernel does not write are copied
        compare(c.batch_host_view(), oracle256[1][:n])
        c.batch_device(d[0].data_ptr(), n, W, H)
        hv = c.batch_host_view()  # no prefetch at all: the view is the compact one, there when the batch ends
        assert hv["kps"] is None and hv["levels"] is None
        compare(hv, oracle256[0][:n], full=False)
        c.set_pipelined_batches(True)
        for i in range(9):  # streaming: mark batch i, read batch i - 1
            c.batch_device(d[i & 1].data_ptr(), n, W, H)
            c.batch_prefetch(compact=True)
            if i:
                compare(c.batch_host_view(previous=1), oracle256[(i - 1) & 1][:n], full=False)
        for back in (0, 1, 2, 3):
            compare(c.batch_host_view(previous=back), oracle256[(8 - back) & 1][:n], full=False)
        c.set_host_results(False)
        c.batch_device(d[1].data_ptr(), n, W, H)
        c.batch_prefetch(compact=True)
        compare(c.batch_host_view(), oracle256[1][:n], full=False)


def test_adaptive_first_pass_changes_the_tile_rows_not_the_results(pkg, batches256, oracle256):
    """The adaptive mode of the top-rows-first pipeline learns in which row the levels' caps fill and shortens the FAST
    tile rows so that the first pass ends just below (orbx_api.cpp, adapt_tile_rows): after a few batches the tile
    table of a frame has more, shorter tile rows.  Every batch before, during and after the change equals the oracle;
    with the early exit switched off the default tile rows come back, and again with it on the learned ones."""
    import torch

    n = 128
    d = [torch.from_numpy(batches256[k][:n]).cuda() for k in range(2)]
    torch.cuda.synchronize()
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=n, **PK)
    with pkg.Context(p) as c:
        cap = c.plan(W, H)["out_capacity"]
        totals = []
        for i in range(10):
            c.batch_device(d[i & 1].data_ptr(), n, W, H)
            compare(c.batch_fetch(0, n, cap), oracle256[i & 1][:n])
            totals.append(c.fast_tile_counts()[1])
        assert totals[0] in (272 * n, 146 * n)  # the default table of either FAST kernel (test_fast_tile_counts)
        assert totals[-1] > totals[0], totals  # shorter tile rows: more tiles
        learned = totals[-1]
        c.set_pipelined_batches(True)
        for i in range(6):
            c.batch_device(d[i & 1].data_ptr(), n, W, H)
        compare(c.batch_fetch(0, n, cap), oracle256[1][:n])
        c.set_pipelined_batches(False)
        c.set_fast_early_exit(False)  # every tile works: the default rows (smaller halo share)
        c.batch_device(d[0].data_ptr(), n, W, H)
        compare(c.batch_fetch(0, n, cap), oracle256[0][:n])
        assert c.fast_tile_counts()[1] == totals[0]
        c.set_fast_early_exit(True)
        c.batch_device(d[1].data_ptr(), n, W, H)
        compare(c.batch_fetch(0, n, cap), oracle256[1][:n])
        assert c.fast_tile_counts()[1] == learned


def test_config4_1080p_as_a_batch(pkg):
    """BASELINE.json configs[4] through the batched device path: 8 stream-B 1920x1080 frames, 12 levels, 4000 features,
    Harris + NMS (level caps > 512: the three-kernel selection; caps not reached in the top rows: the adaptive mode
    falls back from two passes to one), twice in a row and pipelined, against the oracle."""
    import torch

    pk = dict(nfeatures=4000, nlevels=12, scale_factor=1.2, threshold=20, n=9, nms_window=3, patch_size=31,
              blur_levels=2, blur_kind=0)
    w, h, n = 1920, 1080, 8
    frames = pkg.streams.stream_b(n, h, w)
    refs = oracle_on(frames, pk)
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    p = pkg.default_params("gpu", max_width=w, max_height=h, max_batch=n, **pk)
    with pkg.Context(p) as c:
        cap = c.plan(w, h)["out_capacity"]
        for mode in (2, 1, 0):  # adaptive, always two passes, never
            c.set_top_rows_first(mode)
            for _ in range(2):
                c.batch_device(d.data_ptr(), n, w, h)
            compare(c.batch_fetch(0, n, cap), refs)
        c.set_top_rows_first(2)
        c.set_pipelined_batches(True)
        for _ in range(4):
            c.batch_device(d.data_ptr(), n, w, h)
        c.wait()
        compare(c.batch_host_view(), refs)
        compare(c.batch_host_view(previous=True), refs)
