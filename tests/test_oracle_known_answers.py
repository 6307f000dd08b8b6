"""Pins the CPU oracle (oracle/orb_oracle.c) to the known answers that SURVEY.md
recorded from a run of the reference's own src/orb_cpu.cpp on its own fixture
000000.png (SURVEY.md §7 "Minimum slice", §8(a) rows a5/a6, §8(c) rows "Can the
reference's own implementation ... be compiled" and "Defined behaviour around
D15", §8 level sizes / quotas).  These are the only outputs of the real
reference that exist in this image (OpenCV is absent, so orb_cpu.cpp cannot be
rebuilt here); everything else about the oracle is cross-checked against an
independent numpy restatement in test_oracle_vs_numpy.py.
"""
import hashlib

import numpy as np

import oracle_lib as O


def test_fixture_hashes():
    # SURVEY.md §8(c) "Golden vectors": decoded SHA-256 prefixes of the two PNGs
    assert hashlib.sha256(O.load_kitti(0).tobytes()).hexdigest().startswith("11cb4e13a5aa81ad")
    assert hashlib.sha256(O.load_kitti(1).tobytes()).hexdigest().startswith("e7a0c8a2ea2a7bc0")
    img = O.load_kitti(0)
    assert img.shape == (376, 1241) and img.dtype == np.uint8
    assert abs(float(img.mean()) - 89.0) < 0.5 and abs(float(img.std()) - 70.0) < 0.5


def test_fast_counts_threshold_50():
    img = O.load_kitti(0)
    scores, npre, ncor = O.fast_score(img, 50, 9)
    assert npre == 7331 and ncor == 3855
    kps, total = O.nms(scores, 3, 3000)
    assert total == 1178 and len(kps) == 1178  # §8(a) a6
    assert tuple(kps[0]) == (815, 3) and tuple(kps[-1]) == (25, 366)  # §8(c)


def test_fast_counts_threshold_20_and_cap_row():
    img = O.load_kitti(0)
    scores, npre, ncor = O.fast_score(img, 20, 9)
    assert npre == 27392 and ncor == 14415  # §8(a) a5
    kps, total = O.nms(scores, 3, 3000)
    assert total == 4153 and len(kps) == 3000  # §8(a) a6
    assert kps[-1][1] == 174  # "cap 3000 reached at row 174"


def test_cpu_flavour_defaults_on_000000():
    img = O.load_kitti(0)
    kps, ang, desc, valid = O.detect_and_compute_cpu(img)  # OrientedFASTCPU defaults (orb_cpu.hpp:6)
    assert len(kps) == 1178
    assert int((ang == 0).sum()) == 3  # "3 zero angles (border)"
    d2, v2, nskip, noob = O.brief(img, kps, ang)
    assert np.array_equal(d2, desc) and np.array_equal(v2, valid)
    assert nskip == 3220  # "3220 bits (64 keypoints) are legitimately skipped-to-zero"
    vb = np.unpackbits(valid, axis=1, bitorder="little")
    assert int((vb.sum(1) < 256).sum()) == 6  # "6 of 1178 keypoints are affected" by D15
    # skipped bits are zero in the descriptor
    skipped_kps = 0
    # (a skipped bit is valid-and-zero; count keypoints having any bit whose both boxes were rejected)
    assert noob > 0 and noob <= 125  # 125 out-of-bounds SAMPLES (27 wrap + 98 past the end) in <= 125 bits
    del skipped_kps


def test_level_geometry_matches_survey():
    # K and F level sizes and quotas (computed by the reference's formulas)
    kw = [1241, 1034, 862, 718, 598, 499, 416, 346]
    kh = [376, 313, 261, 218, 181, 151, 126, 105]
    assert [O.level_size(1241, 376, 1.2, l) for l in range(8)] == list(zip(kw, kh))
    assert sum(w * h for w, h in zip(kw, kh)) == 1444097
    assert [O.level_quota(1000, 1.2, 8, l) for l in range(8)] == [217, 180, 150, 125, 104, 87, 72, 60]
    fw = [1920, 1600, 1333, 1111, 926, 772, 643, 536, 447, 372, 310, 258]
    fh = [1080, 900, 750, 625, 521, 434, 362, 301, 251, 209, 174, 145]
    assert [O.level_size(1920, 1080, 1.2, l) for l in range(12)] == list(zip(fw, fh))
    assert sum(w * h for w, h in zip(fw, fh)) == 6700616
    assert [O.level_quota(4000, 1.2, 12, l) for l in range(12)] == [750, 625, 521, 434, 362, 301, 251, 209, 174, 145,
                                                                     121, 101]


def test_pattern_table():
    pat = np.load(O.os.path.join(O.ROOT, "tests", "golden", "pattern_31.npy"))
    assert pat.shape == (256, 4) and pat.min() == -13 and pat.max() == 12
    got = np.ctypeslib.as_array((O.C.c_int8 * 1024).in_dll(O.lib(), "oracle_pattern_31")).reshape(256, 4)
    assert np.array_equal(got, pat)
    assert tuple(pat[0]) == (8, -3, 9, 5) and tuple(pat[-1]) == (-1, -6, 0, -11)
    # the two in-tree copies of the table are identical
    a = open(O.os.path.join(O.ROOT, "oracle", "pattern_31.inc")).read()
    b = open(O.os.path.join(O.ROOT, "visual-odometry-gpu_amd", "csrc", "pattern_31.inc")).read()
    assert a == b
