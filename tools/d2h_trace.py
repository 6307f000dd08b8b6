#!/usr/bin/env python3
"""One streaming-consumer configuration for a timeline trace (development aid): d2h_trace.py B lag before|after"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("visual-odometry-gpu_amd")
import torch
B, lag, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
W, H = 1241, 376
fr = [torch.from_numpy(pkg.streams.stream_a(B, first=k * B)).cuda() for k in range(2)]
p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, nfeatures=1000, nlevels=8, blur_levels=2)
with pkg.Context(p) as c:
    c.set_pipelined_batches(True)
    for i in range(4):
        c.batch_device(fr[i & 1].data_ptr(), B, W, H)
    c.wait()
    for i in range(lag):
        c.batch_device(fr[i & 1].data_ptr(), B, W, H)
        c.batch_prefetch(compact=True)
    t = time.perf_counter()
    for i in range(lag, lag + 12):
        c.batch_device(fr[i & 1].data_ptr(), B, W, H)
        if mode == "before":
            c.batch_prefetch(compact=True)
            hv = c.batch_host_view(previous=lag)
        else:
            hv = c.batch_host_view(previous=lag)
            c.batch_prefetch(compact=True)
    c.wait()
    print("fps", round(12 * B / (time.perf_counter() - t)))
