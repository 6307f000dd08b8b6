#!/usr/bin/env python3
"""Stage times of the whole path by number of pyramid levels (development aid): which levels cost what."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("visual-odometry-gpu_amd")
import torch  # noqa: E402

B, W, H = 256, 1241, 376
frames = torch.from_numpy(pkg.streams.stream_a(B)).cuda()
for nl in (8, 7, 6, 5, 4, 1):
    p = pkg.default_params("gpu", max_width=W, max_height=H, max_batch=B, nfeatures=1000, nlevels=nl, blur_levels=2)
    with pkg.Context(p) as c:
        c.set_fast_early_exit(False)
        plan = c.plan(W, H)
        px = int((plan["level_w"].astype(np.int64) * plan["level_h"]).sum())
        for _ in range(3):
            c.batch_device(frames.data_ptr(), B, W, H)
        c.wait()
        c.enable_stage_timing(1)
        acc = {}
        for _ in range(8):
            c.batch_device(frames.data_ptr(), B, W, H)
            c.wait()
            for k, v in c.last_stage_times().items():
                acc[k] = acc.get(k, 0.0) + v / 8
        print(nl, "levels, px/frame", px, {k: round(v * 1e3, 1) for k, v in acc.items() if k in ("blur", "fast_nms", "describe", "select")})
