"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same inputs.  Bar: bit-exact for keypoints, masks, u8 images and
descriptors; 1e-4 for angles (they are in fact bit-equal) and 1e-4 relative
for Harris responses.

Run on the GPU box with:  python -m pytest tests -m gpu -x -q
"""
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


def synth(seed, h, w, kind="rects"):
    """Seeded synthetic frame with KITTI-like statistics (stream B, simplified)."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    img = np.full((h, w), 89.0)
    img += 25.0 * rng.standard_normal((h, w))
    for _ in range(max(8, h * w // 1500)):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        ww, hh = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        img[y0:y0 + hh, x0:x0 + ww] += rng.uniform(20, 120) * rng.choice([-1, 1])
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def kitti0():
    return O.load_kitti(0)


@pytest.fixture(scope="module")
def kitti1():
    return O.load_kitti(1)


@pytest.fixture(scope="module")
def ctx(pkg):
    p = pkg.default_params("gpu", max_width=1920, max_height=1080, max_batch=4)
    c = pkg.Context(p)
    yield c
    c.close()


# ---------------------------------------------------------------------------
# stage operators


@pytest.mark.parametrize("threshold,n", [(50, 9), (20, 9), (20, 12), (35, 16), (10, 1), (0, 9)])
def test_fast_score_map_kitti(ctx, kitti0, threshold, n):
    ref, _, ncor = O.fast_score(kitti0, threshold, n)
    got = ctx.fast_score(kitti0, threshold, n)
    assert np.array_equal(got, ref)
    assert int((got > 0).sum()) <= ncor  # corners with score 0 cannot exist for threshold > 0
    if threshold > 0:
        assert int((got > 0).sum()) == ncor


@pytest.mark.parametrize("shape", [(8, 8), (9, 71), (64, 64), (65, 129), (120, 160), (33, 700), (200, 63)])
def test_fast_score_map_shapes(ctx, shape):
    for kind in ("rects", "noise"):
        img = synth(hash(shape) % 1000, shape[0], shape[1], kind)
        ref, _, _ = O.fast_score(img, 20, 9)
        assert np.array_equal(ctx.fast_score(img, 20, 9), ref), (shape, kind)


@pytest.mark.parametrize("threshold,nms_window,nfeatures", [(50, 3, 3000), (20, 3, 3000), (20, 3, 434), (20, 0, 5000),
                                                            (20, 5, 3000), (20, 7, 3000), (50, 3, 0), (50, 3, 1)])
def test_fast_detect_kitti(ctx, kitti0, threshold, nms_window, nfeatures):
    """Fast() == OrientedFASTCPU::detect: ordered, capped keypoints bit-exact."""
    scores, _, _ = O.fast_score(kitti0, threshold, 9)
    ref, tot = O.nms(scores, nms_window, nfeatures)
    got, gtot = ctx.fast(kitti0, threshold, 9, nms_window, nfeatures)
    assert gtot == tot
    assert np.array_equal(got, ref)


def test_nms_stage_on_score_map(ctx, kitti1):
    scores, _, _ = O.fast_score(kitti1, 20, 9)
    for win in (3, 5):
        ref, tot = O.nms(scores, win, 100000)
        got, gtot = ctx.nms(scores, win, 100000, 0.0)
        assert gtot == tot and np.array_equal(got, ref)


@pytest.mark.parametrize("patch", [9, 31, 1, 41])
def test_orientations(ctx, kitti0, patch):
    kps = O.fast_detect(kitti0, 20, 9, 3, 3000)
    ref = O.orientations(kitti0, kps, patch)
    got = ctx.orientations(kitti0, kps, patch)
    assert np.allclose(got, ref, atol=1e-4, rtol=0)
    # the restated glibc atan2f makes them bit-identical, not merely close
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert int((ref == 0).sum()) == int((got == 0).sum())


def test_brief_given_angles(ctx, kitti0):
    kps = O.fast_detect(kitti0, 50, 9, 3, 3000)
    ang = O.orientations(kitti0, kps, 9)
    ref, valid, nskip, noob = O.brief(kitti0, kps, ang)
    got = ctx.brief(kitti0, kps, ang)
    assert nskip == 3220
    assert np.array_equal(got & valid, ref & valid)
    # the zero-extension rule makes even the D15 bits agree with the restatement
    assert np.array_equal(got, ref)


def test_brief_random_angles_and_border_keypoints(ctx):
    img = synth(7, 90, 130, "noise")
    rng = np.random.default_rng(3)
    kps = np.stack([rng.integers(0, 130, 600), rng.integers(0, 90, 600)], 1).astype(np.int32)
    ang = rng.uniform(-np.pi, np.pi, 600).astype(np.float32)
    ang[:8] = [0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, np.pi / 4, 1e-30, -0.0]
    ref, valid, _, _ = O.brief(img, kps, ang)
    got = ctx.brief(img, kps, ang)
    assert np.array_equal(got, ref)


def test_harris(ctx, kitti0):
    kps = O.fast_detect(kitti0, 20, 9, 3, 1500)
    ref = O.harris(kitti0, kps, 7, 0.04)
    got = ctx.harris(kitti0, kps, 7, 0.04)
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-2)
    assert np.array_equal(got, ref)  # same operation order, no contraction: identical
    # near-border keypoints exercise REFLECT_101 of the Sobel taps
    edge = np.array([[0, 0], [1, 1], [3, 3], [1240, 375], [1237, 372], [0, 200], [1240, 5]], np.int32)
    assert np.array_equal(ctx.harris(kitti0, edge, 7, 0.04), O.harris(kitti0, edge, 7, 0.04))
    for win in (3, 5):
        assert np.array_equal(ctx.harris(kitti0, kps[:200], win, 0.06), O.harris(kitti0, kps[:200], win, 0.06))


@pytest.mark.parametrize("shape", [(376, 1241), (8, 8), (17, 65), (64, 64), (100, 130), (33, 257)])
def test_blur_stages(ctx, kitti0, shape):
    img = kitti0 if shape == (376, 1241) else synth(shape[1], shape[0], shape[1], "noise")
    assert np.array_equal(ctx.blur5_sep(img), O.blur5_sep(img))
    assert np.array_equal(ctx.blur5_273(img), O.blur5_273(img))


def test_conv_sobel_gaussian(ctx, pkg, kitti0):
    img = kitti0[:200, :333].copy()
    for K in (3, 5, 7):
        assert np.array_equal(pkg.orbx.gaussian_kernel(K), O.gaussian_kernel(K))
        assert np.array_equal(ctx.gaussian_blur_conv(img, K), O.gaussian_blur_conv(img, K))
    for d in (0, 1):
        assert np.array_equal(ctx.sobel(img, d), O.sobel_u8(img, d))
    kern = np.random.default_rng(0).uniform(-0.2, 0.4, (5, 5)).astype(np.float32)
    assert np.array_equal(ctx.conv2d(img, kern), O.conv2d_u8(img, kern))


def test_select_top(ctx):
    rng = np.random.default_rng(5)
    r = rng.standard_normal(1000).astype(np.float32)
    r[100:130] = r[5]  # ties must resolve by index
    for keep in (0, 1, 217, 1000, 2000):
        assert np.array_equal(ctx.select_top(r, keep), O.select_top(r, keep))


# ---------------------------------------------------------------------------
# pyramid


@pytest.mark.parametrize("blur_levels,blur_kind", [(0, 0), (1, 0), (2, 0), (2, 1)])
def test_pyramid_levels(pkg, kitti0, blur_levels, blur_kind):
    p = pkg.default_params("gpu", max_width=1241, max_height=376, blur_levels=blur_levels, blur_kind=blur_kind)
    op = O.gpu_params(blur_levels=blur_levels, blur_kind=blur_kind)
    with pkg.Context(p) as c:
        pl = c.plan(1241, 376)
        assert list(pl["level_w"]) == [1241, 1034, 862, 718, 598, 499, 416, 346]
        assert list(pl["level_h"]) == [376, 313, 261, 218, 181, 151, 126, 105]
        for l in range(8):
            assert np.array_equal(c.build_pyramid_level(kitti0, l), O.build_level(kitti0, op, l)), l


# ---------------------------------------------------------------------------
# whole path


def check_whole(got, ref):
    assert got["count"] == len(ref["kps"])
    assert np.array_equal(got["kps"], ref["kps"])
    assert np.array_equal(got["kps_level"], ref["kps_level"])
    assert np.array_equal(got["levels"], ref["levels"])
    assert np.allclose(got["angles"], ref["angles"], atol=1e-4, rtol=0)
    assert np.allclose(got["responses"], ref["responses"], rtol=1e-4, atol=1e-2)
    assert np.array_equal(got["desc"] & ref["valid"], ref["desc"] & ref["valid"])


def test_cpu_flavour_config0(pkg, kitti0):
    """BASELINE.json configs[0]: orb_cpu.cpp detectAndCompute on 000000.png, one level."""
    p = pkg.default_params("cpu", max_width=1241, max_height=376)
    with pkg.Context(p) as c:
        got = c.detect_and_compute(kitti0)
    kps, ang, desc, valid = O.detect_and_compute_cpu(kitti0)
    assert got["count"] == 1178 == len(kps)  # SURVEY.md §7 known answer
    assert np.array_equal(got["kps"], kps)
    assert tuple(got["kps"][0]) == (815, 3) and tuple(got["kps"][-1]) == (25, 366)
    assert np.array_equal(got["angles"].view(np.uint32), ang.view(np.uint32))
    assert int((got["angles"] == 0).sum()) == 3
    assert np.array_equal(got["desc"] & valid, desc & valid)
    assert np.array_equal(got["desc"], desc)
    assert np.all(got["levels"] == 0) and np.all(got["responses"] == 0)


@pytest.mark.parametrize("blur_levels", [0, 2])
def test_gpu_flavour_config1(pkg, kitti0, kitti1, blur_levels):
    """BASELINE.json configs[1]: 1241x376, 8 levels, scale 1.2, 1000 features."""
    p = pkg.default_params("gpu", nfeatures=1000, max_width=1241, max_height=376, blur_levels=blur_levels)
    op = O.gpu_params(nfeatures=1000, blur_levels=blur_levels)
    with pkg.Context(p) as c:
        pl = c.plan(1241, 376)
        assert list(pl["quota"]) == [217, 180, 150, 125, 104, 87, 72, 60]
        for img in (kitti0, kitti1):
            check_whole(c.detect_and_compute(img), O.detect_and_compute_gpu(img, op))


def test_gpu_flavour_config4_1080p(pkg):
    """BASELINE.json configs[4]: 1920x1080, 12 levels, 4000 keypoints, Harris + NMS."""
    img = synth(11, 1080, 1920)
    p = pkg.default_params("gpu", nfeatures=4000, nlevels=12, max_width=1920, max_height=1080, blur_levels=2)
    op = O.gpu_params(nfeatures=4000, nlevels=12, blur_levels=2)
    with pkg.Context(p) as c:
        pl = c.plan(1920, 1080)
        assert list(pl["quota"]) == [750, 625, 521, 434, 362, 301, 251, 209, 174, 145, 121, 101]
        check_whole(c.detect_and_compute(img), O.detect_and_compute_gpu(img, op))


@pytest.mark.parametrize("kw", [dict(threshold=20, n=12), dict(threshold=35, nms_window=5), dict(nms_window=0, nfeatures=200),
                                dict(patch_size=9, nlevels=3, scale_factor=1.5), dict(blur_levels=1), dict(blur_levels=2, blur_kind=1),
                                dict(harris_window=5, harris_k=0.06)])
def test_gpu_flavour_param_sweep(pkg, kw):
    img = synth(21, 120, 160)
    base = dict(nfeatures=300, nlevels=6)
    base.update(kw)
    p = pkg.default_params("gpu", max_width=160, max_height=120, **base)
    with pkg.Context(p) as c:
        check_whole(c.detect_and_compute(img), O.detect_and_compute_gpu(img, O.gpu_params(**base)))


@pytest.mark.parametrize("nms_window", [0, 3, 5, 7])
def test_threshold_zero_on_flat_and_saturated_content(pkg, nms_window):
    """Threshold 0: a flat neighbourhood passes the pre-test and the arc test with score 0, and a score of 0 is no
    keypoint (src/orb_cpu.cpp:110 skips scores <= 0 before it looks at the NMS radius) -- the case a fuzz campaign
    of round 3 caught in the streaming FAST kernel (it kept such corners).  Flat stripes, saturated blocks with one
    grey level of noise, and enough candidates per tile row to overflow the candidate queue."""
    rng = np.random.default_rng(5)
    img = np.zeros((134, 267), np.uint8)
    img[:40] = 77  # flat
    img[40:90] = np.where(rng.random((50, 267)) > 0.5, 255, 0)
    img[90:] = (np.arange(267)[None, :] // 16 % 2 * 255 + rng.integers(0, 2, (44, 267))).clip(0, 255)
    kw = dict(nfeatures=400, nlevels=5, scale_factor=1.5, threshold=0, n=9, nms_window=nms_window, blur_levels=1, blur_kind=1)
    p = pkg.default_params("gpu", max_width=267, max_height=134, **kw)
    with pkg.Context(p) as c:
        check_whole(c.detect_and_compute(img), O.detect_and_compute_gpu(img, O.gpu_params(**kw)))


def test_batch_matches_single_and_device_path(pkg, kitti0, kitti1):
    """Batched device-resident path == per-frame host path == oracle; ragged frame content."""
    import torch

    frames = np.stack([kitti0, kitti1, np.zeros_like(kitti0), np.roll(kitti0, (5, 9), (0, 1))])
    p = pkg.default_params("gpu", nfeatures=1000, max_width=1241, max_height=376, max_batch=4, blur_levels=2)
    op = O.gpu_params(nfeatures=1000, blur_levels=2)
    with pkg.Context(p) as c:
        cap = c.plan(1241, 376)["out_capacity"]
        t = torch.from_numpy(frames).cuda()
        torch.cuda.synchronize()
        c.batch_device(t.data_ptr(), 4, 1241, 376)
        dev = c.batch_fetch(0, 4, cap)
        c.batch_host(frames)
        host = c.batch_fetch(0, 4, cap)
        for k in ("counts", "kps", "angles", "desc", "levels", "responses"):
            assert np.array_equal(dev[k], host[k]), k
        assert dev["counts"][2] == 0  # empty frame: no keypoints, no crash
        for i in range(4):
            ref = O.detect_and_compute_gpu(frames[i], op)
            n = int(dev["counts"][i])
            got = dict(count=n, kps=dev["kps"][i, :n], kps_level=dev["kps_level"][i, :n], levels=dev["levels"][i, :n],
                       angles=dev["angles"][i, :n], responses=dev["responses"][i, :n], desc=dev["desc"][i, :n])
            check_whole(got, ref)


def test_frame_size_change_and_capacity_error(pkg, kitti0):
    p = pkg.default_params("gpu", nfeatures=500, max_width=1241, max_height=376)
    op = O.gpu_params(nfeatures=500)
    with pkg.Context(p) as c:
        for sl in ((slice(0, 376), slice(0, 1241)), (slice(10, 200), slice(100, 900)), (slice(0, 376), slice(0, 1241))):
            img = np.ascontiguousarray(kitti0[sl])
            check_whole(c.detect_and_compute(img), O.detect_and_compute_gpu(img, op))
        got = c.detect_and_compute(kitti0, capacity=10)
        assert got["status"] == pkg.orbx.ERR_CAPACITY and got["count"] > 10 and len(got["kps"]) == 10
        with pytest.raises(pkg.OrbxError):
            c.detect_and_compute(np.zeros((400, 1300), np.uint8))  # larger than max_width/max_height


def test_error_paths(pkg):
    with pytest.raises(pkg.OrbxError):
        pkg.Context(pkg.default_params("gpu", nlevels=40))
    with pytest.raises(pkg.OrbxError):
        pkg.Context(pkg.default_params("gpu", n=17))
    with pytest.raises(pkg.OrbxError):  # level 15 of a 64x64 frame is < 8x8
        pkg.Context(pkg.default_params("gpu", nlevels=16, max_width=64, max_height=64))
    with pkg.Context(pkg.default_params("gpu", max_width=64, max_height=64)) as c:
        with pytest.raises(pkg.OrbxError):  # diagnostics of the last batch: there is none yet
            c.pyramid_pixel_counts()
        with pytest.raises(pkg.OrbxError):
            c.fast_tile_counts()
        for bad in (-1, 3):
            with pytest.raises(pkg.OrbxError):
                c.set_top_rows_first(bad)
        for ok in (0, 1, 2):
            c.set_top_rows_first(ok)
        c.batch_host(np.zeros((1, 64, 64), np.uint8))  # a batch too small for two passes: every pixel produced
        done, total = c.pyramid_pixel_counts()
        assert done == total > 0


def test_fast_early_exit_is_invisible(pkg, kitti0, kitti1):
    """Tiles that provably cannot reach the first `cap` row-major survivors exit early in the
    batched path; outputs must be identical with the early exit on and off, and equal the oracle."""
    frames = np.stack([kitti0, kitti1, np.roll(kitti1, (7, 3), (0, 1)), kitti0[::-1].copy()])
    p = pkg.default_params("gpu", nfeatures=600, max_width=1241, max_height=376, max_batch=4, blur_levels=2)
    with pkg.Context(p) as c:
        cap = c.plan(1241, 376)["out_capacity"]
        c.set_fast_early_exit(True)
        c.batch_host(frames)
        a = c.batch_fetch(0, 4, cap)
        c.set_fast_early_exit(False)
        c.batch_host(frames)
        b = c.batch_fetch(0, 4, cap)
        for k in ("counts", "kps", "kps_level", "angles", "desc", "levels", "responses"):
            assert np.array_equal(a[k], b[k]), k
    ref = O.detect_and_compute_gpu(frames[3], O.gpu_params(nfeatures=600, blur_levels=2))
    n = int(a["counts"][3])
    assert n == len(ref["kps"]) and np.array_equal(a["kps"][3, :n], ref["kps"])
    # CPU flavour: the row-major cap (3000 at threshold 20 is hit at row 174) is exact under early exit
    p = pkg.default_params("cpu", threshold=20, max_width=1241, max_height=376, max_batch=2)
    with pkg.Context(p) as c:
        c.batch_host(np.stack([kitti0, kitti1]))
        r = c.batch_fetch(0, 2, 3000)
    kps, ang, desc, valid = O.detect_and_compute_cpu(kitti0, threshold=20)
    assert r["counts"][0] == 3000 == len(kps) and np.array_equal(r["kps"][0], kps)
    assert np.array_equal(r["desc"][0] & valid, desc & valid)


@pytest.mark.parametrize("w", [8, 9, 10, 11, 63, 64, 65, 243, 244, 245, 246, 247, 248, 249, 250, 251, 252, 253, 254, 255, 256,
                               257, 258, 259, 260, 261, 492, 495, 496, 497, 500, 508, 509, 511, 512, 513, 514, 516, 744, 745,
                               767, 768, 769, 770, 772])
def test_blur_width_sweep(ctx, w):
    """Streaming-blur strip / lane boundaries: the dword holding the last pixel falls on every lane
    position (incl. lanes 0 and 63 of a 256-pixel strip, whose outer neighbours are fetched from memory
    rather than from a lane) and every byte position; heights around the 64-row band and the
    5-row group boundaries."""
    for h in (8, 17, 64, 65, 66, 69, 70, 71, 129):
        img = synth(w * 131 + h, h, w, "noise")
        assert np.array_equal(ctx.blur5_sep(img), O.blur5_sep(img)), (w, h)


@pytest.fixture(scope="module")
def ctx_blur16(pkg):
    """a context whose stand-alone blur is k_blur4 (16 pixels per lane, four row bands per wave): ORBX_BLUR_IMPL=3,
    read when the context is created"""
    import os

    old = os.environ.get("ORBX_BLUR_IMPL")
    os.environ["ORBX_BLUR_IMPL"] = "3"
    try:
        c = pkg.Context(pkg.default_params("gpu", max_width=1920, max_height=1080, max_batch=4))
    finally:
        if old is None:
            del os.environ["ORBX_BLUR_IMPL"]
        else:
            os.environ["ORBX_BLUR_IMPL"] = old
    yield c
    c.close()


@pytest.mark.parametrize("w", [8, 9, 15, 16, 17, 63, 64, 65, 240, 243, 244, 245, 248, 249, 252, 253, 254, 255, 256, 257, 258, 259, 260,
                               261, 262, 272, 497, 508, 509, 511, 512, 513, 514, 516, 767, 768, 769, 770, 772, 1241, 1920])
def test_blur16_width_sweep(ctx_blur16, w):
    """k_blur4: the dword holding the last pixel on every dword of a lane and in the halo of a band's last lane, every
    byte position; heights around the four-band split (a last band that is short, or missing) and the 5-row groups;
    levels taller than one wave's four bands."""
    for h in (8, 9, 17, 21, 64, 66, 71, 129, 385, 391):
        if w > 1000 and h not in (8, 129, 391):
            continue
        img = synth(w * 131 + h, h, w, "noise")
        assert np.array_equal(ctx_blur16.blur5_sep(img), O.blur5_sep(img)), (w, h)


@pytest.mark.parametrize("shape", [(64, 64), (65, 65), (127, 129), (128, 128), (129, 127), (70, 200), (200, 70),
                                   (191, 255), (193, 257)])
def test_whole_path_tile_boundaries(pkg, shape):
    """Frame sizes around the 64x64 FAST tiles, the 256x16 pyramid tiles and the 8-byte resize
    windows (right-edge clamp), multi-level, Harris selection, blur on."""
    h, w = shape
    img = synth(h * 7 + w, h, w)
    kw = dict(nfeatures=400, nlevels=4, blur_levels=2)
    p = pkg.default_params("gpu", max_width=w, max_height=h, **kw)
    op = O.gpu_params(**kw)
    with pkg.Context(p) as c:
        for l in range(4):
            assert np.array_equal(c.build_pyramid_level(img, l), O.build_level(img, op, l)), (shape, l)
        check_whole(c.detect_and_compute(img), O.detect_and_compute_gpu(img, op))


def test_scale_factors_and_resize_paths(pkg, kitti1):
    """scale factors on both sides of the 8-byte-window limit (scale <= 2 per level)."""
    img = np.ascontiguousarray(kitti1[:300, :1000])
    for sf, nl in ((1.1, 6), (1.5, 5), (2.0, 4), (2.5, 3), (3.0, 3)):
        kw = dict(nfeatures=500, nlevels=nl, scale_factor=sf)
        p = pkg.default_params("gpu", max_width=1000, max_height=300, **kw)
        op = O.gpu_params(**kw)
        with pkg.Context(p) as c:
            for l in range(nl):
                assert np.array_equal(c.build_pyramid_level(img, l), O.build_level(img, op, l)), (sf, l)
            check_whole(c.detect_and_compute(img), O.detect_and_compute_gpu(img, op))


@pytest.mark.parametrize("w", [1241, 1022, 643, 600, 250])
def test_staged_resize_levels(pkg, w):
    """Levels of scale 2..3.2 through the batched path: k_pyrblur stages the strip's span of the two source rows in
    LDS (spans that end with the source row for every width mod 4 -- the frame's buffer descriptor zeroes a dword
    that straddles the frame's last byte --, spans of fewer and of more than 64 8-byte chunks, one and several
    strips per level, a single-strip level that must fall back to the 2-byte gathers).  Noise frames and per-level
    caps above the number of corners of the staged level: every pixel of it matters."""
    h = 200
    for sf in (2.1, 2.6, 3.1):
        kw = dict(nfeatures=2800, nlevels=3, scale_factor=sf, threshold=25, blur_levels=2)
        p = pkg.default_params("gpu", max_width=w, max_height=h, max_batch=2, **kw)
        op = O.gpu_params(**kw)
        frames = np.stack([synth(31 * w + int(10 * sf), h, w, "noise"), synth(w + int(10 * sf), h, w, "rects")])
        with pkg.Context(p) as c:
            cap = c.plan(w, h)["out_capacity"]
            c.batch_host(frames)
            r = c.batch_fetch(0, 2, cap)
            for i in range(2):
                ref = O.detect_and_compute_gpu(frames[i], op)
                n = int(r["counts"][i])
                got = dict(count=n, kps=r["kps"][i, :n], kps_level=r["kps_level"][i, :n], levels=r["levels"][i, :n],
                           angles=r["angles"][i, :n], responses=r["responses"][i, :n], desc=r["desc"][i, :n])
                check_whole(got, ref)
                lv1 = got["levels"] == 1
                assert 15 <= lv1.sum() < c.plan(w, h)["quota"][1], (w, sf, lv1.sum())  # the level has corners, none cut


def test_random_frames_many(pkg):
    """A few hundred keypoint-rich random frames through the batched path vs the oracle (rare-event hunting:
    rounding ties in the rotation, ties in NMS and in the Harris ranking)."""
    kw = dict(nfeatures=800, nlevels=5, threshold=12, blur_levels=2)
    h, w = 120, 200
    p = pkg.default_params("gpu", max_width=w, max_height=h, max_batch=32, **kw)
    op = O.gpu_params(**kw)
    with pkg.Context(p) as c:
        cap = c.plan(w, h)["out_capacity"]
        for rnd in range(4):
            frames = np.stack([synth(1000 * rnd + i, h, w, "rects" if i % 3 else "noise") for i in range(32)])
            c.batch_host(frames)
            r = c.batch_fetch(0, 32, cap)
            for i in range(32):
                ref = O.detect_and_compute_gpu(frames[i], op)
                n = int(r["counts"][i])
                got = dict(count=n, kps=r["kps"][i, :n], kps_level=r["kps_level"][i, :n], levels=r["levels"][i, :n],
                           angles=r["angles"][i, :n], responses=r["responses"][i, :n], desc=r["desc"][i, :n])
                check_whole(got, ref)
                assert np.array_equal(got["angles"].view(np.uint32), ref["angles"].view(np.uint32))
                assert np.array_equal(got["responses"].view(np.uint32), ref["responses"].view(np.uint32))


def test_fast_candidate_queue_overflow(ctx):
    """More than F2_QCAP (2048) pre-test survivors in one 64x64 tile: the queue-less path."""
    img = synth(5, 130, 200, "noise")
    for t in (1, 2, 5):
        ref, npre, _ = O.fast_score(img, t, 9)
        assert npre > 3 * 2048  # really overflows (several tiles)
        assert np.array_equal(ctx.fast_score(img, t, 9), ref), t
        s, _, _ = O.fast_score(img, t, 9)
        kr, tot = O.nms(s, 3, 100000)
        kg, gtot = ctx.fast(img, t, 9, 3, 100000)
        assert gtot == tot and np.array_equal(kg, kr)


@pytest.mark.parametrize("select_mode", [0, 1])
def test_zero_feature_budget(pkg, select_mode):
    """nfeatures too small for any per-level quota (orb.cpp:62 truncates to 0): count must be 0, not stale memory
    (regression: the describe launch that writes the counts is skipped when the output capacity is 0)."""
    img = synth(3, 131, 55, "noise")
    p = pkg.default_params("gpu", max_width=55, max_height=131, max_batch=4, nfeatures=0, nlevels=2)
    p.select_mode = select_mode
    with pkg.Context(p) as c:
        c.batch_host(np.stack([img] * 4))
        r = c.batch_fetch(0, 4, 1)
        assert np.array_equal(r["counts"], np.zeros(4, np.int32))
        assert c.detect_and_compute(img)["count"] == 0


def test_fast_tile_counts(pkg, kitti0, kitti1, fast_impl):
    """orbx_fast_tile_counts: with the early exit off every tile works; with it on, fewer do, and the
    results are the same (checked bit for bit in test_fast_early_exit_is_invisible)."""
    frames = np.stack([kitti0, kitti1] * 4)
    p = pkg.default_params("gpu", max_width=1241, max_height=376, max_batch=8, nfeatures=1000, blur_levels=2)
    with pkg.Context(p) as c:
        c.set_fast_early_exit(False)
        c.batch_host(frames)
        w0, t0 = c.fast_tile_counts()
        plan = c.plan(1241, 376)
        tile_h = 47  # orbx_fast3_tile_h(1): 7 x 7 rows of the score region - 2 x NMS radius
        if fast_impl == "3":  # tiles of 128 x <= 47
            cols = [-(-int(w) // 128) for w in plan["level_w"]]
            want = 272
        else:
            # units of the streaming kernel: strips of 64 dwords, the outer dword of a side is halo (62 productive;
            # the image's own borders need none) x tile rows of <= 47 rows
            cols = [max(1, -(-(-(-int(w) // 4) - 2) // 62)) for w in plan["level_w"]]
            want = 146
        per_frame = sum(s * -(-int(h) // tile_h) for s, h in zip(cols, plan["level_h"]))
        assert w0 == t0 and t0 == 8 * per_frame == 8 * want
        row0 = sum(cols)  # tile row 0 of every level
        c.set_fast_early_exit(True)
        c.batch_host(frames)
        w1, t1 = c.fast_tile_counts()
        assert t1 == t0 and 8 * row0 <= w1 <= t0  # at least tile row 0 of every level; how many
        # of the others exit depends on how far their dispatch trails the completion of the rows above


def test_graph_and_plain_launch_paths_agree(pkg, kitti0, kitti1):
    """run_batch replays a captured hipGraph per batch shape; with stage timing on it takes the plain
    launch path.  Same results either way, also when batch shapes alternate (graph cache)."""
    frames = np.stack([kitti0, kitti1, kitti1, kitti0])
    p = pkg.default_params("gpu", max_width=1241, max_height=376, max_batch=4, nfeatures=700, blur_levels=2)
    with pkg.Context(p) as c:
        cap = c.plan(1241, 376)["out_capacity"]

        def run(n):
            c.batch_host(frames[:n])
            r = c.batch_fetch(0, n, cap)
            return [r[k].copy() for k in ("counts", "kps", "angles", "responses", "desc", "levels")]

        ref4, ref1 = run(4), run(1)            # graph path (captured twice)
        for _ in range(3):                     # alternating shapes: served from the graph cache
            for n, ref in ((1, ref1), (4, ref4), (2, None), (3, None)):
                got = run(n)
                if ref is not None:
                    assert all(np.array_equal(a, b) for a, b in zip(got, ref))
        c.enable_stage_timing(1)               # plain path
        got4, got1 = run(4), run(1)
        c.enable_stage_timing(0)
        assert all(np.array_equal(a, b) for a, b in zip(got4, ref4))
        assert all(np.array_equal(a, b) for a, b in zip(got1, ref1))
        t = c.last_stage_times()
        assert t["total"] > 0


def test_pipelined_result_fetch(pkg, kitti0, kitti1):
    """Two result blocks: the D2H copy of batch i (orbx_batch_prefetch, copy stream) overlaps the kernels
    of batch i+1; orbx_batch_fetch_previous then returns batch i.  Same results as the blocking fetch."""
    import torch

    batches = [np.stack([kitti0, kitti1]), np.stack([kitti1, np.roll(kitti0, (3, 7), (0, 1))]),
               np.stack([kitti0[::-1].copy(), kitti1[:, ::-1].copy()]), np.stack([kitti1, kitti1])]
    p = pkg.default_params("gpu", max_width=1241, max_height=376, max_batch=2, nfeatures=800, blur_levels=2)
    keys = ("counts", "kps", "kps_level", "angles", "responses", "desc", "levels")
    with pkg.Context(p) as c:
        cap = c.plan(1241, 376)["out_capacity"]
        dev = [torch.from_numpy(b).cuda() for b in batches]
        torch.cuda.synchronize()
        want = []
        for d in dev:  # blocking reference
            c.batch_device(d.data_ptr(), 2, 1241, 376)
            want.append(c.batch_fetch(0, 2, cap))
        for rounds in range(2):  # second round: the graph cache serves both result blocks
            got = []
            c.batch_device(dev[0].data_ptr(), 2, 1241, 376)
            c.batch_prefetch()
            for i in range(1, len(dev)):
                c.batch_device(dev[i].data_ptr(), 2, 1241, 376)
                got.append(c.batch_fetch(0, 2, cap, previous=True))
                c.batch_prefetch()
            got.append(c.batch_fetch(0, 2, cap))
            for g, w in zip(got, want):
                for k in keys:
                    assert np.array_equal(g[k], w[k]), k
            # zero-copy views of the pinned mirrors: the last batch and the one before it
            hv, hp = c.batch_host_view(), c.batch_host_view(previous=True)
            for v_, w in ((hv, want[-1]), (hp, want[-2])):
                assert np.array_equal(v_["counts"], w["counts"])
                for f_ in range(2):
                    n_ = int(w["counts"][f_])
                    for k in keys[1:]:
                        assert np.array_equal(v_[k][f_, :n_], w[k][f_, :n_]), k
        with pytest.raises(pkg.OrbxError):
            pkg.Context(p).batch_fetch(0, 1, cap, previous=True)  # nothing has run yet


@pytest.mark.parametrize("shape", [(376, 1241), (120, 160), (64, 250), (65, 497), (200, 70), (33, 257), (90, 745)])
def test_fused_pyramid_blur_equals_separate_kernels(pkg, shape):
    """Blur on every level: the batched path builds and blurs the pyramid in one kernel (k_pyrblur); with
    orbx_set_fused_pyramid_blur(0) the two kernels run separately.  Same results, equal to the oracle
    (strip / lane / band boundaries of both kernels, all three resize modes of scale > 2: 8 levels reach scale 3.6)."""
    h, w = shape
    img = synth(h * 7 + w, h, w, "noise" if min(h, w) < 100 else "rects")
    nl = 8 if min(h, w) >= 60 else 4
    kw = dict(nfeatures=600, nlevels=nl, blur_levels=2)
    p = pkg.default_params("gpu", max_width=w, max_height=h, max_batch=3, **kw)
    frames = np.stack([img, img[::-1].copy(), np.roll(img, 5, 1)])
    with pkg.Context(p) as c:
        cap = max(c.plan(w, h)["out_capacity"], 1)
        c.batch_host(frames)
        a = c.batch_fetch(0, 3, cap)
        c.set_fused_pyramid_blur(False)
        c.batch_host(frames)
        b = c.batch_fetch(0, 3, cap)
        for k in ("counts", "kps", "kps_level", "angles", "desc", "levels", "responses"):
            assert np.array_equal(a[k], b[k]), k
    ref = O.detect_and_compute_gpu(frames[1], O.gpu_params(**kw))
    n = int(a["counts"][1])
    assert n == len(ref["kps"]) and np.array_equal(a["kps"][1, :n], ref["kps"])
    assert np.array_equal(a["desc"][1, :n], ref["desc"])
