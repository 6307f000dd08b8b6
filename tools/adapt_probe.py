#!/usr/bin/env python3
"""The adaptive first pass on other content (development aid): frames/s and tile-table changes per top-rows mode."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("visual-odometry-gpu_amd")
import numpy as np
import torch
B, W, H = 256, 1241, 376
streams = {
    "A (benchmark)": [pkg.streams.stream_a(B, first=k * B) for k in range(4)],
    "B (denser, 1241x376)": [pkg.streams.stream_b(B, H, W) for k in range(1)],
}
# half of every frame dark at the top: the caps fill far lower
dark = [b.copy() for b in streams["A (benchmark)"][:2]]
for b in dark:
    b[:, :150, :] //= 8
streams["A, top 150 rows dark"] = dark
# alternating regimes every 6 batches
streams["A / A-dark alternating every 6 batches"] = None
for name, bl in streams.items():
    for mode in (2, 1, 0):
        p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, nfeatures=1000, nlevels=8, blur_levels=2)
        with pkg.Context(p) as c:
            c.set_top_rows_first(mode)
            c.set_pipelined_batches(True)
            if bl is None:
                seq = [streams["A (benchmark)"][0]] * 6 + [dark[0]] * 6
            else:
                seq = bl
            d = {id(x): torch.from_numpy(x).cuda() for x in seq}
            torch.cuda.synchronize()
            totals = []
            for i in range(12):
                c.batch_device(d[id(seq[i % len(seq)])].data_ptr(), B, W, H)
                c.wait()
            t = time.perf_counter()
            n = 96
            for i in range(n):
                c.batch_device(d[id(seq[i % len(seq)])].data_ptr(), B, W, H)
                if i % 8 == 7:
                    c.wait()
                    totals.append(c.fast_tile_counts()[1] // B)
            c.wait()
            fps = n * B / (time.perf_counter() - t)
            print("%-42s mode %d: %7.0f frames/s, units per frame over time %s, pixels produced %.3f" % (
                name, mode, fps, sorted(set(totals)), c.pyramid_pixel_counts()[0] / max(c.pyramid_pixel_counts()[1], 1)), flush=True)
