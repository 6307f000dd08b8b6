"""The LK oracle (oracle/lk_oracle.c) restates OpenCV's calcOpticalFlowPyrLK, which is absent here
(PARITY UNPINNED): these tests pin what can be pinned without it -- the published definitions of
pyrDown / Scharr on small hand-checkable inputs (independent numpy restatement) and the tracker's
behaviour on synthetic motion with a known answer."""
import numpy as np

import oracle_lib as O


def smooth_image(seed, h, w):
    """Band-limited texture evaluated analytically, so that sub-pixel shifts are exact."""
    rng = np.random.default_rng(seed)
    k = 24
    fx, fy = rng.uniform(0.02, 0.25, k), rng.uniform(0.02, 0.25, k)
    ph, amp = rng.uniform(0, 2 * np.pi, k), rng.uniform(0.3, 1.0, k)

    def f(dx, dy):
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        v = np.zeros((h, w))
        for i in range(k):
            v += amp[i] * np.sin(fx[i] * (x + dx) + fy[i] * (y + dy) + ph[i])
        v = 128 + 100 * v / np.abs(v).max()
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return f


def np_pyr_down(img):
    h, w = img.shape
    k = np.array([1, 4, 6, 4, 1])
    p = np.pad(img.astype(np.int64), 2, mode="reflect")
    dh, dw = (h + 1) // 2, (w + 1) // 2
    out = np.zeros((dh, dw), np.int64)
    for i in range(5):
        for j in range(5):
            out += k[i] * k[j] * p[i:i + 2 * dh:2, j:j + 2 * dw:2][:dh, :dw]
    return ((out + 128) >> 8).astype(np.uint8)


def np_scharr(img):
    p = np.pad(img.astype(np.int64), 1, mode="reflect")
    h, w = img.shape
    s = lambda dy, dx: p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    dx = 3 * (s(-1, 1) - s(-1, -1)) + 10 * (s(0, 1) - s(0, -1)) + 3 * (s(1, 1) - s(1, -1))
    dy = 3 * (s(1, -1) - s(-1, -1)) + 10 * (s(1, 0) - s(-1, 0)) + 3 * (s(1, 1) - s(-1, 1))
    return np.stack([dx, dy], -1).astype(np.int16)


def test_pyr_down_and_scharr_vs_numpy():
    rng = np.random.default_rng(0)
    for shape in ((31, 47), (64, 64), (33, 2 * 21 + 3), (376, 1241)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        assert np.array_equal(O.lk_pyr_down(img), np_pyr_down(img)), shape
        assert np.array_equal(O.lk_scharr(img), np_scharr(img)), shape
    flat = np.full((40, 50), 77, np.uint8)
    assert np.all(O.lk_pyr_down(flat) == 77) and not O.lk_scharr(flat).any()
    ramp = np.tile(np.arange(60, dtype=np.uint8) * 3, (40, 1))
    d = O.lk_scharr(ramp)
    assert np.all(d[2:-2, 2:-2, 0] == 16 * 2 * 3) and not d[2:-2, 2:-2, 1].any()  # (3+10+3) * (I[x+1]-I[x-1])


def test_tracker_recovers_known_translation():
    f = smooth_image(1, 240, 320)
    prev = f(0, 0)
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(40, 280, 200), rng.uniform(40, 200, 200)], 1).astype(np.float32)
    for dx, dy in ((0.0, 0.0), (1.25, -0.5), (-3.6, 2.3), (7.5, 5.25), (-13.0, 9.0)):
        nxt = f(-dx, -dy)  # content moves by (+dx, +dy)
        out, st, err, top = O.lk_track(prev, nxt, pts)
        assert top == 3
        assert st.mean() > 0.97, (dx, dy, st.mean())
        e = np.abs(out[st == 1] - pts[st == 1] - np.float32([dx, dy]))
        assert np.percentile(e, 95) < 0.15, (dx, dy, np.percentile(e, 95))
        assert np.median(err[st == 1]) < 2.0


def test_tracker_status_and_levels():
    f = smooth_image(3, 120, 160)
    prev, nxt = f(0, 0), f(-2.0, 0.0)
    pts = np.float32([[80, 60], [-40, 60], [400, 60], [80, -50], [159, 119], [0, 0]])
    out, st, err, top = O.lk_track(prev, nxt, pts)
    assert top == 2  # 160x120 -> 80x60 -> 40x30 -> (20x15 is not larger than the 21x21 window)
    assert st[0] == 1 and abs(out[0, 0] - 82) < 0.1 and abs(out[0, 1] - 60) < 0.1
    assert st[1] == 0 and st[2] == 0 and st[3] == 0  # window entirely outside the image
    # a textureless pair: the minimum-eigenvalue test rejects every point
    flat = np.full((120, 160), 90, np.uint8)
    out, st, err, _ = O.lk_track(flat, flat, pts[:1])
    assert st[0] == 0
    # zero iterations: the point is only propagated through the levels
    out, st, err, _ = O.lk_track(prev, nxt, pts[:1], max_iters=0)
    assert st[0] == 1 and np.array_equal(out[0], pts[0])
    # n = 0
    out, st, err, _ = O.lk_track(prev, nxt, np.zeros((0, 2), np.float32))
    assert len(out) == 0
