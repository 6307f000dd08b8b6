"""Synthetic input streams of the benchmark and the tests.  Plain data generation: nothing here
touches liborbx or the oracle.  The two KITTI frames are the reference's own data files
(`images/000000.png`, `000001.png`), kept as fixtures under tests/golden/."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_kitti(i=0):
    """Decoded pixels of the reference's KITTI frame i (0 or 1): 376 x 1241 uint8."""
    return np.load(os.path.join(ROOT, "tests", "golden", "kitti_%06d.npz" % i))["image"]


def stream_a(n, first=0):
    """Stream A: frames derived deterministically from the two KITTI fixtures (roll + small Gaussian noise), KITTI
    statistics kept.  Frame i depends on i only."""
    base = [load_kitti(0), load_kitti(1)]
    out = np.empty((n,) + base[0].shape, np.uint8)
    for j in range(n):
        i = first + j
        rng = np.random.default_rng(1000 + i)
        img = np.roll(base[i & 1], ((3 * i) % 17, (5 * i) % 11), (0, 1)).astype(np.int16)
        img += np.rint(rng.normal(0.0, 2.0, img.shape)).astype(np.int16)
        out[j] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def stream_a_device(torch, first, n, device):
    """The same recipe generated on the GPU (a per-frame seeded torch generator instead of numpy's): 8000 frames
    take ~1 s instead of ~80 s of host time.  Frame i depends on i only, so any sharding of the stream sees the
    same frames (they differ from stream_a's: another noise generator)."""
    base = [torch.from_numpy(load_kitti(k)).to(device).to(torch.int16) for k in (0, 1)]
    out = torch.empty((n,) + tuple(base[0].shape), dtype=torch.uint8, device=device)
    g = torch.Generator(device=device)
    for j in range(n):
        i = first + j
        g.manual_seed(1000 + i)
        noise = torch.round(torch.randn(base[0].shape, generator=g, device=device) * 2.0).to(torch.int16)
        img = torch.roll(base[i & 1], shifts=((3 * i) % 17, (5 * i) % 11), dims=(0, 1)) + noise
        out[j] = img.clamp_(0, 255).to(torch.uint8)
    return out


def stream_b(n, h, w, first=0):
    """Stream B: synthetic frames of any resolution (noise field + 400 random rectangles)."""
    out = np.empty((n, h, w), np.uint8)
    for j in range(n):
        rng = np.random.default_rng(first + j)
        img = 89.0 + 30.0 * rng.standard_normal((h, w))
        for _ in range(400):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            ww, hh = int(rng.integers(4, 120)), int(rng.integers(4, 120))
            img[y0:y0 + hh, x0:x0 + ww] += rng.uniform(20, 120) * rng.choice([-1, 1])
        out[j] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out
