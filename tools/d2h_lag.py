#!/usr/bin/env python3
"""Streaming consumer with the results on the host (development aid): frames/s by how many batches the consumer lags."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("visual-odometry-gpu_amd")
import torch
B, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1241, 376
fr = [torch.from_numpy(pkg.streams.stream_a(B, first=k * B)).cuda() for k in range(2)]
p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, nfeatures=1000, nlevels=8, blur_levels=2)
with pkg.Context(p) as c:
    c.set_pipelined_batches(True)
    for i in range(4):
        c.batch_device(fr[i & 1].data_ptr(), B, W, H)
    c.wait()
    t = time.perf_counter()
    for i in range(20):
        c.batch_device(fr[i & 1].data_ptr(), B, W, H)
    c.wait()
    print("no copy", round(20 * B / (time.perf_counter() - t)), flush=True)
    for hostres, lag, mode in ((0, 1, "after"), (0, 1, "before"), (0, 2, "after"), (0, 2, "before"),
                               (1, 1, "after"), (1, 1, "before"), (1, 2, "after"), (1, 2, "before"), (1, 3, "before")):
        c.wait()
        c.set_host_results(bool(hostres))
        for _ in range(1):
            c.wait()
            for i in range(lag):
                c.batch_device(fr[i & 1].data_ptr(), B, W, H)
                c.batch_prefetch(compact=True)
            t = time.perf_counter()
            tot = 0
            for i in range(lag, lag + 20):
                c.batch_device(fr[i & 1].data_ptr(), B, W, H)
                if mode == "before":
                    c.batch_prefetch(compact=True)
                    hv = c.batch_host_view(previous=lag)
                else:
                    hv = c.batch_host_view(previous=lag)
                    c.batch_prefetch(compact=True)
                tot += int(hv["counts"][0])
            c.wait()
            print("host_results", hostres, "lag", lag, "prefetch", mode, "the view:", round(20 * B / (time.perf_counter() - t)), flush=True)
