"""End to end over every row of SURVEY.md §8: examples/vo_frontend.cpp is the image-side half of the
reference's tracking loop (VisualOdom::run, feature_tracking.cpp:44-126) on liborbx -- PNG frames of a KITTI
sequence directory -> ORB detect -> pyramidal LK tracking, with the ORB + 2-NN + ratio-test fallback when
fewer than 150 tracks survive.  A synthetic sequence with known motion checks the chain."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from test_kitti_io import write_png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "vo_frontend.bin")


def test_example_builds():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "vo_frontend.bin"], stdout=subprocess.DEVNULL)
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_tracking_loop_on_a_synthetic_sequence(tmp_path):
    assert os.path.exists(BIN), "build() must have produced tests/cpp/vo_frontend.bin"
    seq = tmp_path / "data_odometry_gray" / "dataset" / "sequences" / "00" / "image_0"
    seq.mkdir(parents=True)
    base = O.load_kitti(0)
    shifts = [(0, 0), (3, 1), (6, 2), (9, 3), (12, 4)]  # content moves by (+3, +1) px per frame
    for i, (dx, dy) in enumerate(shifts):
        write_png(seq / ("%06d.png" % i), np.roll(base, (dy, dx), (0, 1)), 0, (0, 1, 2), idat_split=2)

    def run(nfeatures):
        r = subprocess.run([BIN, str(tmp_path), "00", "100", str(nfeatures)], capture_output=True, text=True, timeout=300)
        print(r.stdout, r.stderr)
        assert r.returncode == 0, r.stdout + r.stderr
        rows = [ln.split() for ln in r.stdout.splitlines() if ln and ln[0].isdigit()]
        assert len(rows) == len(shifts) and rows[0][2] == "detect"
        return rows

    # cv::ORB::create(3000): every frame is tracked by LK
    rows = run(3000)
    assert int(rows[0][1]) > 2000
    for row in rows[1:]:
        n, how, fx, fy = int(row[1]), row[2], float(row[3]), float(row[4])
        assert how == "track" and n > 1000, row
        assert abs(fx - 3) < 0.2 and abs(fy - 1) < 0.2, row
    # a detector budget below the reference's 150-track threshold: every frame falls back to
    # ORB + 2-NN + ratio test (feature_tracking.cpp:70-72)
    rows = run(140)
    for row in rows[1:]:
        n, how, fx, fy = int(row[1]), row[2], float(row[3]), float(row[4])
        assert how == "match" and n > 30, row
        assert abs(fx - 3) <= 1.0 and abs(fy - 1) <= 1.0, row  # median flow; keypoints are integer pixels
