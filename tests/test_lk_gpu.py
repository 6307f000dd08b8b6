"""Pyramidal Lucas-Kanade tracker: HIP path vs the CPU oracle, bit for bit
(positions compared as float bit patterns, status and error exactly), through the C ABI."""
import numpy as np
import pytest

import oracle_lib as O
from test_lk_oracle import smooth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    p = pkg.default_params("gpu", max_width=64, max_height=64, max_batch=1)
    with pkg.Context(p) as c:
        yield c


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def check(ctx, prev, nxt, pts, **kw):
    ro, rs, re, _ = O.lk_track(prev, nxt, pts, **kw)
    go, gs, ge = ctx.lk_track(prev, nxt, pts, **kw)
    assert np.array_equal(gs, rs)
    assert same(go, ro), np.abs(go - ro).max()
    assert same(ge, re)
    return go, gs, ge


def test_kitti_pair_reference_parameters(ctx):
    """The reference's own call: 21x21 window, 3 levels above the base, 30 iterations / 0.01 (feature_tracking.cpp:175-181)
    on the two KITTI fixtures, from FAST corners of the first one."""
    a, b = O.load_kitti(0), O.load_kitti(1)
    kps = O.fast_detect(a, 20, 9, 3, 3000).astype(np.float32)
    out, st, err = check(ctx, a, b, kps)
    assert st.mean() > 0.8
    # prev=None reuses the device pyramid of the previous `next` (img1 = img2.clone()): tracking b -> a
    sel = out[st == 1]
    go, gs, ge = ctx.lk_track(None, a, sel)
    ro, rs, re, _ = O.lk_track(b, a, sel)
    assert np.array_equal(gs, rs) and same(go, ro) and same(ge, re)
    # forward-backward consistency of the tracks that survive both ways
    fb = np.abs(go[gs == 1] - kps[st == 1][gs == 1]).max(1)
    assert np.median(fb) < 0.1


@pytest.mark.parametrize("win,max_level,max_iters,eps", [(21, 3, 30, 0.01), (5, 0, 10, 0.03), (31, 5, 3, 0.001),
                                                        (9, 2, 100, 0.0), (15, 7, 0, 0.01), (3, 1, 30, 0.01)])
def test_parameter_sweep_with_border_points(ctx, win, max_level, max_iters, eps):
    f = smooth_image(5, 150, 211)
    prev, nxt = f(0, 0), f(-2.4, 1.7)
    rng = np.random.default_rng(win)
    pts = np.concatenate([
        np.stack([rng.uniform(-30, 240, 300), rng.uniform(-30, 180, 300)], 1),   # includes points outside the image
        np.float32([[0, 0], [210, 149], [0.5, 148.5], [105.25, 74.75], [-21, 10], [211, 75], [1e4, 1e4], [-1e4, 3]]),
    ]).astype(np.float32)
    check(ctx, prev, nxt, pts, win=win, max_level=max_level, max_iters=max_iters, epsilon=eps)


def test_noise_and_flat_images(ctx):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    b = np.roll(a, (1, 2), (0, 1))
    pts = np.stack([rng.uniform(0, 131, 200), rng.uniform(0, 97, 200)], 1).astype(np.float32)
    check(ctx, a, b, pts)
    flat = np.full((97, 131), 200, np.uint8)
    out, st, err = check(ctx, flat, flat, pts)
    assert not st.any()
    check(ctx, a, b, np.zeros((0, 2), np.float32))


def test_size_sweep_and_errors(ctx, pkg):
    for (h, w) in ((23, 23), (22, 64), (64, 22), (45, 300), (376, 1241), (1080, 1920)):
        f = smooth_image(h + w, h, w)
        prev, nxt = f(0, 0), f(-1.5, 0.75)
        rng = np.random.default_rng(h)
        pts = np.stack([rng.uniform(0, w, 150), rng.uniform(0, h, 150)], 1).astype(np.float32)
        check(ctx, prev, nxt, pts)
    with pytest.raises(pkg.OrbxError):
        ctx.lk_track(prev, nxt, pts, win=33)
    with pytest.raises(pkg.OrbxError):
        ctx.lk_track(prev, nxt, pts, max_level=8)
    small = np.zeros((10, 10), np.uint8)
    with pytest.raises(pkg.OrbxError):
        ctx.lk_track(None, small, pts[:1])  # no previous call of that geometry


def test_min_eigenvalue_borderline_case(ctx):
    """Found by tools/fuzz_lk.py (seed 5, iteration 53734): a window hanging over the left border whose
    gradient matrix is almost singular -- the minimum-eigenvalue test then hinges on the last bit of the square
    root (the device's native sqrt is 1 ulp off; the kernel must use the correctly rounded one)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_min_eig_case.npz"))
    check(ctx, z["a"], z["b"], z["pts"], win=14, max_level=0, max_iters=31, epsilon=0.0)
