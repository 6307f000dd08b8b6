"""Cross-checks the C oracle against an INDEPENDENT numpy restatement written
from the reference's Python prototypes (orb.py:4-94 FAST/NMS/orientation,
blur.py:4-31 the /273 blur) and from the kernel arithmetic of src/cuda/*.cu.
Two independently written restatements agreeing bit for bit is what backs the
oracle where no output of the real reference exists (see orb_oracle.h).
CPU only; sized to run in well under a minute.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

RING = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0),
        (-3, -1), (-2, -2), (-1, -3)]  # (dx, dy), orb_cpu.cpp:8-13 / orb.py:8-13

_libm = C.CDLL("libm.so.6")
_libm.cosf.restype = C.c_float
_libm.cosf.argtypes = [C.c_float]
_libm.sinf.restype = C.c_float
_libm.sinf.argtypes = [C.c_float]
_libm.atan2f.restype = C.c_float
_libm.atan2f.argtypes = [C.c_float, C.c_float]


def synth(seed, h, w):
    rng = np.random.default_rng(seed)
    img = 89.0 + 25.0 * rng.standard_normal((h, w))
    for _ in range(max(6, h * w // 1500)):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        img[y0:y0 + int(rng.integers(3, 30)), x0:x0 + int(rng.integers(3, 30))] += rng.uniform(20, 120) * rng.choice([-1, 1])
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ---- numpy restatements ----------------------------------------------------

def np_fast_score(img, t, n):
    """orb.py:17-55 vectorised: pre-test, circular n-run test, score = sum |Ip - ring|."""
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # 16 x H' x W'
    br = ring >= c + t
    dk = ring <= c - t
    pre_b = br[[0, 4, 8, 12]].sum(0)
    pre_d = (dk[[0, 4, 8, 12]] & ~br[[0, 4, 8, 12]]).sum(0)  # else-if in orb_cpu.cpp:51-54
    pre = np.maximum(pre_b, pre_d) >= 3
    corner = np.zeros_like(pre)
    for i in range(16):
        idx = [(i + j) % 16 for j in range(n)]
        corner |= br[idx].all(0) | dk[idx].all(0)
    corner &= pre
    score = np.abs(ring - c).sum(0)
    out = np.zeros((h, w), np.float32)
    out[3:h - 3, 3:w - 3] = np.where(corner, score, 0)
    return out, int(pre.sum()), int(corner.sum())


def np_nms(scores, win, cap):
    """orb.py:57-66 (window max equality, row-major order) + the C++ cap."""
    h, w = scores.shape
    r = win // 2
    keep = scores > 0
    keep[:3] = keep[-3:] = False
    keep[:, :3] = keep[:, -3:] = False
    if r:
        mx = np.zeros_like(scores)
        pad = np.pad(scores, r)
        for dy in range(2 * r + 1):
            for dx in range(2 * r + 1):
                mx = np.maximum(mx, pad[dy:dy + h, dx:dx + w])
        keep &= scores == mx
    ys, xs = np.nonzero(keep)  # row-major
    kps = np.stack([xs, ys], 1).astype(np.int32)
    return kps[:cap], len(kps)


def np_orientations(img, kps, patch):
    """orb.py:70-94: intensity centroid; 0 when the patch leaves the image."""
    h, w = img.shape
    pr = patch // 2
    I = img.astype(np.int64)
    d = np.arange(-pr, pr + 1)
    out = np.zeros(len(kps), np.float32)
    for i, (x, y) in enumerate(kps):
        if x - pr < 0 or x + pr >= w or y - pr < 0 or y + pr >= h:
            continue
        p = I[y - pr:y + pr + 1, x - pr:x + pr + 1]
        m10 = int((p * d[None, :]).sum())
        m01 = int((p * d[:, None]).sum())
        out[i] = _libm.atan2f(float(m01), float(m10))
    return out


def np_brief(img, kps, angles, pat):
    """orb_cpu.cpp:203-258 with the zero-extension rule of orb_oracle.h."""
    h, w = img.shape
    pad = np.zeros((h + 48, w + 48), np.int64)
    pad[24:24 + h, 24:24 + w] = img
    integ = pad.cumsum(0).cumsum(1)
    integ = np.pad(integ, ((1, 0), (1, 0)))

    def box(cx, cy):  # 5x5 sum centred (cx, cy) of the zero-extended image
        x0, y0, x1, y1 = cx - 2 + 24, cy - 2 + 24, cx + 3 + 24, cy + 3 + 24
        return integ[y1, x1] + integ[y0, x0] - integ[y0, x1] - integ[y1, x0]

    f32 = np.float32
    desc = np.zeros((len(kps), 32), np.uint8)
    for k, ((kx, ky), a) in enumerate(zip(kps, angles)):
        c, s = f32(_libm.cosf(float(a))), f32(_libm.sinf(float(a)))
        x1, y1, x2, y2 = (pat[:, j].astype(np.float32) for j in range(4))

        def lround(v):
            return (np.sign(v) * np.floor(np.abs(v.astype(np.float64)) + 0.5)).astype(np.int64)

        dx1, dy1 = lround(c * x1 - s * y1), lround(s * x1 + c * y1)
        dx2, dy2 = lround(c * x2 - s * y2), lround(s * x2 + c * y2)
        cx1, cy1, cx2, cy2 = kx + dx1, ky + dy1, kx + dx2, ky + dy2
        skip = (cx1 < 2) | (cy1 < 2) | (cx1 > w - 1) | (cy1 > h - 1) | (cx2 < 2) | (cy2 < 2) | (cx2 > w - 1) | (cy2 > h - 1)
        bits = (box(cx1, cy1) < box(cx2, cy2)) & ~skip
        desc[k] = np.packbits(bits.astype(np.uint8), bitorder="little")
    return desc


def np_reflect(n, r):
    idx = np.arange(-r, n + r)
    idx = np.where(idx < 0, -idx, idx)
    return np.where(idx >= n, 2 * n - idx - 2, idx)


def np_blur_sep(img):
    """GaussianBlur1D.cu: exact rational arithmetic, round-half-even of S/256."""
    h, w = img.shape
    p = img.astype(np.int64)[np_reflect(h, 2)][:, np_reflect(w, 2)]
    k = np.array([1, 4, 6, 4, 1])
    hp = sum(k[i] * p[:, i:i + w] for i in range(5))
    S = sum(k[i] * hp[i:i + h] for i in range(5))
    q, rem = S // 256, S % 256
    return (q + ((rem > 128) | ((rem == 128) & (q % 2 == 1)))).astype(np.uint8)


def np_blur_273(img):
    """blur.py:4-20 kernel, with the CUDA wrapper's round-half-even (GaussianBlur.cu:129)."""
    h, w = img.shape
    p = img.astype(np.float64)[np_reflect(h, 2)][:, np_reflect(w, 2)]
    k = np.array([[1, 4, 7, 4, 1], [4, 16, 26, 16, 4], [7, 26, 41, 26, 7], [4, 16, 26, 16, 4], [1, 4, 7, 4, 1]], np.float64)
    S = sum(k[i, j] * p[i:i + h, j:j + w] for i in range(5) for j in range(5))
    return np.rint(S / 273.0).astype(np.uint8)


def np_resize(img, dw, dh):
    """OpenCV 4.x generic 8UC1 INTER_LINEAR, vectorised."""
    sh, sw = img.shape

    def taps(dn, sn, clamp_frac):
        scale = 1.0 / (np.float64(dn) / sn)
        f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        if clamp_frac:
            lo, hi = s < 0, s >= sn - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, sn - 1, s))
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = taps(dw, sw, True)
    sy, b0, b1 = taps(dh, sh, False)
    I = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    rows = I[:, sx] * a0 + I[:, sx1] * a1  # sh x dw
    r0 = rows[np.clip(sy, 0, sh - 1)]
    r1 = rows[np.clip(sy + 1, 0, sh - 1)]
    return ((((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)


def np_harris(img, kps, K, kk):
    """Full-frame version of the HarrisScore.cu intent, float32, (i,j) accumulation order."""
    h, w = img.shape
    f = np.float32
    p = img.astype(f)[np_reflect(h, 1)][:, np_reflect(w, 1)]
    gx = (p[0:h, 2:] + f(2) * p[1:h + 1, 2:] + p[2:, 2:]) - (p[0:h, 0:w] + f(2) * p[1:h + 1, 0:w] + p[2:, 0:w])
    gy = (p[2:, 0:w] + f(2) * p[2:, 1:w + 1] + p[2:, 2:]) - (p[0:h, 0:w] + f(2) * p[0:h, 1:w + 1] + p[0:h, 2:])
    g = O.gaussian_kernel(K)
    r = K // 2
    ry, rx = np_reflect(h, r), np_reflect(w, r)

    def blur(prod):
        pp = prod[ry][:, rx]
        acc = np.zeros((h, w), f)
        for i in range(K):
            for j in range(K):
                acc = acc + pp[i:i + h, j:j + w] * g[i, j]
        return acc

    A, B, Cc = blur(gx * gx), blur(gx * gy), blur(gy * gy)
    R = (A * Cc - B * B) - (f(kk) * (A + Cc)) * (A + Cc)
    return R[kps[:, 1], kps[:, 0]]


# ---- tests -------------------------------------------------------------------

@pytest.mark.parametrize("t,n", [(50, 9), (20, 9), (20, 12), (5, 16), (30, 1)])
def test_fast_score_kitti(t, n):
    img = O.load_kitti(0)
    ref, npre, ncor = O.fast_score(img, t, n)
    got, gpre, gcor = np_fast_score(img, t, n)
    assert (gpre, gcor) == (npre, ncor)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("win,cap", [(3, 3000), (3, 100), (0, 100000), (5, 100000), (7, 100000)])
def test_nms_kitti(win, cap):
    scores, _, _ = O.fast_score(O.load_kitti(1), 20, 9)
    ref, tot = O.nms(scores, win, cap)
    got, gtot = np_nms(scores, win, cap)
    assert gtot == tot and np.array_equal(got, ref)


@pytest.mark.parametrize("seed", range(4))
def test_fast_nms_small_random(seed):
    img = synth(seed, 40 + 7 * seed, 50 + 11 * seed)
    ref, _, _ = O.fast_score(img, 15, 9)
    got, _, _ = np_fast_score(img, 15, 9)
    assert np.array_equal(got, ref)
    assert np.array_equal(np_nms(got, 3, 10 ** 6)[0], O.nms(ref, 3, 10 ** 6)[0])
    assert np.array_equal(O.fast_detect(img, 15, 9, 3, 50), np_nms(got, 3, 50)[0])


@pytest.mark.parametrize("patch", [9, 31])
def test_orientations(patch):
    img = O.load_kitti(0)
    kps = O.fast_detect(img, 50, 9, 3, 3000)
    ref = O.orientations(img, kps, patch)
    got = np_orientations(img, kps, patch)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_brief():
    img = O.load_kitti(0)
    pat = np.load(O.os.path.join(O.ROOT, "tests", "golden", "pattern_31.npy"))
    kps = O.fast_detect(img, 50, 9, 3, 3000)[::3]
    ang = O.orientations(img, kps, 9)
    ref, valid, _, _ = O.brief(img, kps, ang)
    got = np_brief(img, kps, ang, pat)
    assert np.array_equal(got, ref)
    # integral-image identity used by the reference: cv::integral layout
    ii = np.zeros((img.shape[0] + 1, img.shape[1] + 1), np.int32)
    O.lib().oracle_integral(img.ctypes.data_as(O.u8p), img.shape[1], img.shape[0], img.shape[1], ii.ctypes.data_as(O.i32p))
    assert np.array_equal(ii[1:, 1:], img.astype(np.int64).cumsum(0).cumsum(1)) and not ii[0].any() and not ii[:, 0].any()


@pytest.mark.parametrize("shape", [(376, 1241), (8, 8), (9, 13), (40, 64), (65, 33)])
def test_blur(shape):
    img = O.load_kitti(0) if shape == (376, 1241) else np.random.default_rng(shape[1]).integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(O.blur5_sep(img), np_blur_sep(img))
    assert np.array_equal(O.blur5_273(img), np_blur_273(img))


def test_resize_levels():
    img = O.load_kitti(0)
    for l in (1, 2, 5, 7):
        w, h = O.level_size(1241, 376, 1.2, l)
        assert np.array_equal(O.resize_linear(img, w, h), np_resize(img, w, h)), l
    small = np.random.default_rng(0).integers(0, 256, (37, 53), dtype=np.uint8)
    for dw, dh in ((53, 37), (20, 11), (8, 8), (100, 80)):
        assert np.array_equal(O.resize_linear(small, dw, dh), np_resize(small, dw, dh))
    assert np.array_equal(O.resize_linear(small, 53, 37), small)  # identity scale reproduces the source


def test_conv_and_sobel_wrappers():
    img = O.load_kitti(1)[:120, :200].copy()
    for d, k in ((0, [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]), (1, [[-1, -2, -1], [0, 0, 0], [1, 2, 1]])):
        p = img.astype(np.int64)[np_reflect(120, 1)][:, np_reflect(200, 1)]
        acc = sum(k[i][j] * p[i:i + 120, j:j + 200] for i in range(3) for j in range(3))
        assert np.array_equal(O.sobel_u8(img, d), np.clip(acc, 0, 255).astype(np.uint8))  # D4: negatives saturate to 0
    g = O.gaussian_kernel(7)
    assert abs(float(g.sum()) - 1.0) < 1e-6 and g[3, 3] == g.max() and np.allclose(g, g.T)
    sig = np.float32(0.3) * np.float32(3.0) + np.float32(0.8)  # GaussianBlur.cpp:15 -> 1.7
    assert abs(float(g[3, 4] / g[3, 3]) - float(np.exp(-1.0 / (2 * sig * sig)))) < 1e-6


def test_harris_full_frame_equals_pointwise():
    img = O.load_kitti(0)[:150, :260].copy()
    kps = O.fast_detect(img, 20, 9, 3, 500)
    edge = np.array([[0, 0], [259, 149], [1, 148], [3, 3]], np.int32)
    for K, kk, pts in ((7, 0.04, kps), (5, 0.06, kps[:50]), (7, 0.04, edge)):
        ref = O.harris(img, pts, K, kk)
        got = np_harris(img, pts, K, kk)
        assert np.array_equal(got, ref)


def test_select_top_is_stable_sort():
    rng = np.random.default_rng(1)
    r = rng.standard_normal(500).astype(np.float32)
    r[40:60] = r[3]
    order = np.lexsort((np.arange(500), -r.astype(np.float64)))
    for keep in (0, 1, 100, 500, 900):
        assert np.array_equal(O.select_top(r, keep), order[:min(keep, 500)].astype(np.int32))


def test_orchestrator_is_composition_of_stages():
    """oracle_detect_and_compute_gpu == the stage functions chained as orb.cpp:58-109 does."""
    img = O.load_kitti(1)
    p = O.gpu_params(nfeatures=600, blur_levels=1)
    res = O.detect_and_compute_gpu(img, p)
    off = 0
    for l in range(8):
        lvl = O.build_level(img, p, l)
        if l == 0:
            assert np.array_equal(lvl, img)
        else:
            w, h = O.level_size(1241, 376, 1.2, l)
            assert np.array_equal(lvl, O.blur5_sep(O.resize_linear(img, w, h)))
        q = O.level_quota(600, 1.2, 8, l)
        cand = O.fast_detect(lvl, 20, 9, 3, 2 * q)
        R = O.harris(lvl, cand, 7, 0.04)
        sel = O.select_top(R, q)
        kl = cand[sel]
        n = len(kl)
        assert np.array_equal(res["kps_level"][off:off + n], kl)
        assert np.array_equal(res["responses"][off:off + n], R[sel])
        assert np.all(np.diff(R[sel]) <= 0)
        ang = O.orientations(lvl, kl, 31)
        assert np.array_equal(res["angles"][off:off + n], ang)
        assert np.array_equal(res["desc"][off:off + n], O.brief(lvl, kl, ang)[0])
        s = np.float32(O.level_scale(1.2, l))
        assert np.array_equal(res["kps"][off:off + n], (kl.astype(np.float32) * s).astype(np.int32))
        assert np.all(res["levels"][off:off + n] == l)
        off += n
    assert off == len(res["kps"])
