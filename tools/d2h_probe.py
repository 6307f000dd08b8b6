#!/usr/bin/env python3
"""Host-link probe (development aid): D2H rate of pinned copies of a compact result block, on one and on two streams."""
import time
import torch
n = 40 * 1024 * 1024
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
for parts in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(parts)]
    step = n // parts
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            for i, s in enumerate(ss):
                with torch.cuda.stream(s):
                    h[i * step:(i + 1) * step].copy_(d[i * step:(i + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print("parts", parts, "GB/s", round(10 * n / dt / 1e9, 2), flush=True)
# a kernel writing straight into pinned memory
hd = h.cuda() if False else None
