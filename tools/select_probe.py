"""Where the fused compaction + Harris + selection kernel spends its time (diagnostics)."""
import importlib, sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
pkg = importlib.import_module("visual-odometry-gpu_amd")
import oracle_lib as O
base = O.load_kitti(0)
B = 64
frames = np.stack([np.roll(base, (i % 7, i % 5), (0, 1)) for i in range(B)])
d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
kw = dict(nfeatures=1000, nlevels=8)
p = pkg.default_params("gpu", max_width=1241, max_height=376, max_batch=B, blur_levels=2, **kw)
with pkg.Context(p) as c:
    for _ in range(3):
        c.batch_device(d.data_ptr(), B, 1241, 376); c.wait()
    c.enable_stage_timing(1)
    acc = {}
    for _ in range(10):
        c.batch_device(d.data_ptr(), B, 1241, 376); c.wait()
        for k, v in c.last_stage_times().items(): acc[k] = acc.get(k, 0) + v * 100
    print(os.environ.get("ORBX_SELECT_ABLATE"), {k: round(v, 1) for k, v in acc.items()}, flush=True)
