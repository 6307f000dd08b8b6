#!/usr/bin/env python3
"""Generate data fixtures from the reference's DATA files (run in the build
container only; /root/reference does not exist on the GPU box).

Outputs (all data, no reference source text):
  tests/golden/kitti_000000.npz, kitti_000001.npz
      decoded 8-bit gray pixels of /root/reference/000000.png, 000001.png
      (decoded SHA-256 prefixes 11cb4e13a5aa81ad /
      e7a0c8a2ea2a7bc0 recorded in SURVEY.md §8c).
  oracle/pattern_31.inc, visual-odometry-gpu_amd/csrc/pattern_31.inc
      the 256x4 learned ORB test-pair table (x1,y1,x2,y2 per bit), i.e. the
      numeric contents of bit_pattern_31_ (src/orb_pattern.cpp:4-260), written
      as a flat comma-separated int8 list, 16 numbers per line.
  tests/golden/kitti_000000.png
      the reference's 000000.png itself (a data file), for the PNG reader.
  tests/golden/matching_orb_{gt_path,est_path,scale}.txt
      the first 120 lines of results/matching_orb/*.txt (data written by the
      reference's savePaths, src/feature_matching.cpp:295-322).
"""
import hashlib
import os
import re
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def decode_png(name):
    from PIL import Image

    im = Image.open(os.path.join(REF, name))
    a = np.array(im)
    assert a.dtype == np.uint8 and a.ndim == 2, (a.dtype, a.shape)
    return a


def main():
    expect = {"000000.png": "11cb4e13a5aa81ad", "000001.png": "e7a0c8a2ea2a7bc0"}
    for name, pre in expect.items():
        a = decode_png(name)
        h = hashlib.sha256(a.tobytes()).hexdigest()
        assert h.startswith(pre), (name, h)
        out = os.path.join(ROOT, "tests", "golden", "kitti_" + name.replace(".png", ".npz"))
        np.savez_compressed(out, image=a, sha256=np.array(h))
        print(name, a.shape, h[:16], "->", out, os.path.getsize(out))

    src = open(os.path.join(REF, "src", "orb_pattern.cpp")).read()
    body = src[src.index("{") + 1 : src.rindex("}")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//.*", "", body)
    vals = [int(t) for t in re.findall(r"-?\d+", body)]
    assert len(vals) == 1024 and min(vals) >= -13 and max(vals) <= 13, (len(vals), min(vals), max(vals))
    lines = []
    for i in range(0, 1024, 16):
        lines.append(",".join(str(v) for v in vals[i : i + 16]) + ",")
    text = (
        "/* 256 BRIEF test pairs (x1,y1,x2,y2), int8; data extracted by tools/make_fixtures.py\n"
        "   from the reference's bit_pattern_31_ table (src/orb_pattern.cpp:4-260). */\n"
        + "\n".join(lines)
        + "\n"
    )
    for rel in ("oracle/pattern_31.inc", "visual-odometry-gpu_amd/csrc/pattern_31.inc"):
        with open(os.path.join(ROOT, rel), "w") as f:
            f.write(text)
    np.save(os.path.join(ROOT, "tests", "golden", "pattern_31.npy"), np.array(vals, dtype=np.int8).reshape(256, 4))
    # I/O formats: one of the reference's own PNG frames, byte for byte (a data
    # file), and the head of the trajectory files one of its runs wrote (results/matching_orb/*.txt,
    # the format metric.py:49-51 reads back)
    import shutil

    shutil.copyfile(os.path.join(REF, "000000.png"), os.path.join(ROOT, "tests", "golden", "kitti_000000.png"))
    for name in ("gt_path.txt", "est_path.txt", "scale.txt"):
        lines = open(os.path.join(REF, "results", "matching_orb", name)).read().splitlines(True)[:120]
        with open(os.path.join(ROOT, "tests", "golden", "matching_orb_" + name), "w") as f:
            f.writelines(lines)
    print("pattern sha256", hashlib.sha256(np.array(vals, dtype=np.int8).tobytes()).hexdigest()[:16])


if __name__ == "__main__":
    sys.exit(main())
