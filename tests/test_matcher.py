"""Next row: Hamming 2-NN + ratio test over 256-bit descriptors
(flann->knnMatch(des1, des2, matches, 2) + `m.distance < 0.8 * n.distance`,
feature_matching.cpp:166-181).  CPU part: the C oracle against a numpy restatement;
GPU part: the HIP kernel against the oracle, bit for bit.  The reference's FLANN LSH
index is approximate and lives in OpenCV (absent): parity with it is UNPINNED; the
restated semantics are exact brute force with ties to the lower train index."""
import numpy as np
import pytest

import oracle_lib as O


def np_knn2(q, t):
    q = np.unpackbits(q.reshape(-1, 32), axis=1).astype(np.int32)
    t = np.unpackbits(t.reshape(-1, 32), axis=1).astype(np.int32)
    idx = np.full((len(q), 2), -1, np.int32)
    dist = np.full((len(q), 2), -1, np.int32)
    if len(t) == 0:
        return idx, dist
    d = q @ (1 - t).T + (1 - q) @ t.T  # Hamming distances, nq x nt
    order = np.argsort(d, axis=1, kind="stable")  # stable: ties keep the lower train index
    for k in range(min(2, len(t))):
        idx[:, k] = order[:, k]
        dist[:, k] = np.take_along_axis(d, order[:, k:k + 1], 1)[:, 0]
    return idx, dist


def rand_desc(seed, n, near=None, flips=20):
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if near is not None and len(near):  # make some descriptors close to `near` ones (realistic matches + ties)
        k = min(n, len(near)) // 2
        d[:k] = near[:k]
        for i in range(k):
            bits = rng.integers(0, 256, rng.integers(0, flips))
            for b in bits:
                d[i, b >> 3] ^= 1 << (b & 7)
    return d


@pytest.mark.parametrize("nq,nt", [(0, 5), (5, 0), (3, 1), (1, 2), (100, 257), (300, 256), (513, 1000)])
def test_oracle_knn2_vs_numpy(nq, nt):
    t = rand_desc(nt, nt)
    q = rand_desc(1000 + nq, nq, near=t)
    if nt >= 4:
        t[3] = t[1]  # exact duplicates: tie on distance, lower index must win
    idx, dist = O.knn2(q, t)
    ridx, rdist = np_knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_oracle_ratio_rule():
    t = rand_desc(1, 400)
    q = rand_desc(2, 300, near=t)
    idx, dist = O.knn2(q, t)
    qi, ti, d1 = O.match_ratio(q, t, 0.8)
    keep = (idx[:, 1] >= 0) & (dist[:, 0].astype(np.float32).astype(np.float64) < 0.8 * dist[:, 1].astype(np.float32).astype(np.float64))
    assert np.array_equal(qi, np.nonzero(keep)[0]) and np.array_equal(ti, idx[keep, 0]) and np.array_equal(d1, dist[keep, 0])
    # with integer distances the double rule is exactly 5*d1 < 4*d2
    assert np.array_equal(keep, (idx[:, 1] >= 0) & (5 * dist[:, 0] < 4 * dist[:, 1]))
    assert 20 < len(qi) < 300


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt", [(0, 5), (5, 0), (3, 1), (1, 2), (100, 257), (300, 256), (513, 1000), (1000, 995)])
def test_gpu_knn2_and_ratio(pkg, nq, nt):
    t = rand_desc(nt, nt)
    q = rand_desc(1000 + nq, nq, near=t)
    if nt >= 4:
        t[3] = t[1]
    with pkg.Context(pkg.default_params("gpu", max_width=64, max_height=64)) as c:
        idx, dist = c.knn2(q, t)
        ridx, rdist = O.knn2(q, t)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
        for ratio in (0.8, 0.7, 1.0):
            got = c.match_ratio(q, t, ratio)
            ref = O.match_ratio(q, t, ratio)
            for a, b in zip(got, ref):
                assert np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_batch_match_consecutive_frames(pkg):
    """The VO loop's shape: detect+describe a batch, match frame i -> i+1 on the device, against the
    oracle chain (oracle ORB on both frames, oracle matcher)."""
    k0, k1 = O.load_kitti(0), O.load_kitti(1)
    frames = np.stack([k0, k1, np.roll(k1, (2, 5), (0, 1)), k0])
    kw = dict(nfeatures=1000)
    p = pkg.default_params("gpu", max_width=1241, max_height=376, max_batch=4, **kw)
    with pkg.Context(p) as c:
        cap = c.plan(1241, 376)["out_capacity"]
        c.batch_host(frames)
        c.batch_match_consecutive(0.8)
        res = c.batch_fetch(0, 4, cap)
        refs = [O.detect_and_compute_gpu(f, O.gpu_params(**kw)) for f in frames]
        for pair in range(3):
            qi, ti, d1 = c.batch_match_fetch(pair, cap)
            rq, rt, rd = O.match_ratio(refs[pair]["desc"], refs[pair + 1]["desc"], 0.8)
            assert np.array_equal(qi, rq) and np.array_equal(ti, rt) and np.array_equal(d1, rd)
            assert len(qi) > 50
        # KITTI 000000 -> 000001: matched keypoints move by a few pixels
        qi, ti, _ = c.batch_match_fetch(0, cap)
        disp = np.hypot(*(res["kps"][0][qi] - res["kps"][1][ti]).T.astype(np.float64))
        assert np.median(disp) < 15
        with pytest.raises(pkg.OrbxError):
            c.batch_match_fetch(3, cap)
