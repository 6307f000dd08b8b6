"""Time the LK tracker call and its parts (diagnostics)."""
import importlib, sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
pkg = importlib.import_module("visual-odometry-gpu_amd")
import oracle_lib as O
a, b = O.load_kitti(0), O.load_kitti(1)
p = pkg.default_params("gpu", max_width=1241, max_height=376, max_batch=1)
with pkg.Context(p) as c:
    kps, _ = c.fast(a, 20, 9, 3, 3000)
    pts = kps.astype(np.float32)
    for n in (0, 1, 100, 1000, 3000):
        c.lk_track(a, b, pts[:n])
        t = time.perf_counter()
        for i in range(20):
            c.lk_track(None, a if i % 2 == 0 else b, pts[:n])
        print(n, "points:", round((time.perf_counter() - t) / 20 * 1e3, 3), "ms per call", flush=True)
    for it in (1, 5, 30):
        t = time.perf_counter()
        for i in range(20):
            c.lk_track(None, a if i % 2 == 0 else b, pts, max_iters=it)
        print("max_iters", it, round((time.perf_counter() - t) / 20 * 1e3, 3), "ms per call", flush=True)
