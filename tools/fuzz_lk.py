"""Random-configuration parity hunt for the LK tracker: HIP path vs oracle/lk_oracle.c, bit for bit.
usage: python tools/fuzz_lk.py [seconds] [seed] [only_iteration]   (the third argument replays one iteration verbosely)"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
pkg = importlib.import_module("visual-odometry-gpu_amd")
import oracle_lib as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
k0 = O.load_kitti(0)
p = pkg.default_params("gpu", max_width=64, max_height=64, max_batch=1, nlevels=1)
t0 = time.time(); it = 0; npts = 0
with pkg.Context(p) as c:
    while time.time() - t0 < budget:
        h, w = int(rng.integers(8, 300)), int(rng.integers(8, 400))
        kind = rng.integers(0, 4)
        if kind == 0:
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:
            y0, x0 = int(rng.integers(0, 376 - h)), int(rng.integers(0, 1241 - w))
            a = np.ascontiguousarray(k0[y0:y0 + h, x0:x0 + w])
        elif kind == 2:
            a = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
            a[rng.integers(0, h, 20), rng.integers(0, w, 20)] = 255
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            a = ((np.sin(xx * 0.2) * np.cos(yy * 0.13) + 1) * 127).astype(np.uint8)
        b = np.roll(a, (int(rng.integers(-4, 5)), int(rng.integers(-4, 5))), (0, 1))
        if rng.random() < 0.3:
            b = np.clip(b.astype(np.int16) + rng.integers(-6, 7, b.shape), 0, 255).astype(np.uint8)
        n = int(rng.integers(0, 300))
        pts = np.stack([rng.uniform(-40, w + 40, n), rng.uniform(-40, h + 40, n)], 1).astype(np.float32)
        kw = dict(win=int(rng.integers(3, 32)), max_level=int(rng.integers(0, 8)), max_iters=int(rng.integers(0, 40)),
                  epsilon=float(rng.choice([0.0, 0.001, 0.01, 0.03, 0.5])))
        if only >= 0 and it != only:
            it += 1
            if it > only:
                break
            continue
        ro, rs, re, _ = O.lk_track(a, b, pts, **kw)
        go, gs, ge = c.lk_track(a, b, pts, **kw)
        ok = np.array_equal(gs, rs) and np.array_equal(go.view(np.uint32), ro.view(np.uint32)) and \
            np.array_equal(ge.view(np.uint32), re.view(np.uint32))
        if not ok:
            bad = np.nonzero((gs != rs) | (go.view(np.uint32) != ro.view(np.uint32)).any(1) | (ge.view(np.uint32) != re.view(np.uint32)))[0]
            for i in bad[:5]:
                print("point", i, pts[i], "gpu", go[i], gs[i], ge[i], "oracle", ro[i], rs[i], re[i])
        assert ok, (it, h, w, kind, kw)
        if only >= 0:
            print("iteration", it, "ok")
            break
        it += 1; npts += n
print("lk fuzz ok: %d configurations, %d points in %.0f s" % (it, npts, time.time() - t0))
