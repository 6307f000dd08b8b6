"""CPU-only checks of the drop-in boundary: liborbx.so loads, exports every
symbol include/orbx.h declares, the POD layouts match the header, and -- with
no GPU -- context creation fails LOUDLY (no CPU fallback exists)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "orbx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(pkg):
    assert header_functions() == sorted(pkg.orbx.EXPORTS)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.orbx.load()
    for name in header_functions():
        assert hasattr(lib, name), name


def test_every_entry_point_cites_the_reference():
    """Each C-ABI function names the reference interface it replaces (file:line)."""
    src = open(os.path.join(ROOT, "include", "orbx.h")).read()
    for cite in ("include/orb.hpp:37", "src/orb.cpp:58-109", "include/Fast.cuh:5", "include/NMS.cuh:5",
                 "include/Brief.cuh:5", "include/HarrisScore.cuh:5", "include/Convolution.cuh:5",
                 "include/GaussianBlur.cuh:3", "include/GaussianBlur.cuh:4", "include/GaussianBlur.hpp:6",
                 "include/Sobel.hpp:6", "src/orb.cpp:111-120", "src/orb_cpu.cpp:271-276"):
        assert cite in src, cite


def test_pod_layouts(pkg):
    o = pkg.orbx
    assert C.sizeof(o.Params) == 16 * 4
    assert o.Params.scale_factor.offset == 4 and o.Params.harris_k.offset == 32 and o.Params.device.offset == 60
    assert C.sizeof(o.BatchView) == 7 * 8 + 8 + 8 and o.BatchView.keypoints16.offset == 64


def test_default_params_are_the_reference_defaults(pkg):
    g = pkg.default_params("gpu")  # orb.hpp:12,36 ; orb.cpp:65
    assert (g.nfeatures, g.nlevels, g.threshold, g.n, g.nms_window, g.patch_size) == (500, 8, 20, 9, 3, 31)
    assert abs(g.scale_factor - 1.2) < 1e-6 and g.harris_window == 7 and abs(g.harris_k - 0.04) < 1e-7
    assert g.select_mode == pkg.orbx.SELECT_HARRIS and g.blur_levels == pkg.orbx.BLUR_NONE
    c = pkg.default_params("cpu")  # orb_cpu.hpp:6
    assert (c.nfeatures, c.threshold, c.n, c.nms_window, c.patch_size, c.nlevels) == (3000, 50, 9, 3, 9, 1)
    assert c.select_mode == pkg.orbx.SELECT_ROWMAJOR


def test_status_strings_and_version(pkg):
    lib = pkg.orbx.load()
    assert lib.orbx_status_string(0) == b"ok"
    assert b"capacity" in lib.orbx_status_string(pkg.orbx.ERR_CAPACITY)
    assert "gfx950" in pkg.orbx.version()


def test_host_side_gaussian_kernel_needs_no_gpu(pkg, oracle):
    import numpy as np

    for K in (3, 5, 7, 9):
        assert np.array_equal(pkg.orbx.gaussian_kernel(K), oracle.gaussian_kernel(K))
    with pytest.raises(pkg.OrbxError):
        pkg.orbx.gaussian_kernel(4)  # "Kernel size must be odd" (GaussianBlur.cpp:8-11)


def test_invalid_params_rejected_before_touching_the_gpu(pkg):
    for kw in (dict(nlevels=0), dict(nlevels=17), dict(n=0), dict(n=17), dict(scale_factor=1.0), dict(nms_window=9),
               dict(patch_size=43), dict(harris_window=4), dict(max_batch=0), dict(max_width=4)):
        with pytest.raises(pkg.OrbxError) as e:
            pkg.Context(pkg.default_params("gpu", **kw))
        assert e.value.status == pkg.orbx.ERR_INVALID_ARG, kw


def test_no_silent_cpu_fallback(pkg):
    """Without a GPU orbx_create must fail with NO_DEVICE (or HIP); with one it must succeed."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(pkg.OrbxError) as e:
        pkg.Context(pkg.default_params("gpu"))
    assert e.value.status in (pkg.orbx.ERR_NO_DEVICE, pkg.orbx.ERR_HIP)


def test_product_never_touches_the_oracle():
    """Nothing under the package (or include/) may reference oracle/."""
    for base in ("visual-odometry-gpu_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".inc")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "oracle_lib" not in txt and "liborb_oracle" not in txt and "orb_oracle.h" not in txt, f


def test_no_result_altering_environment_switch_in_the_shipping_library():
    """A drop-in library must not let an environment variable corrupt its output: the only ORBX_*
    switches compiled into liborbx.so choose between result-identical implementations."""
    blob = open(os.path.join(ROOT, "visual-odometry-gpu_amd", "liborbx.so"), "rb").read()
    # whole C strings only (getenv names); macro names inside error-message text are not switches
    found = set(m.decode() for m in re.findall(rb"\x00(ORBX_[A-Z0-9_]{3,})(?=\x00)", blob))
    allowed = {
        "ORBX_FAST_EARLY",     # 0: every FAST tile does the full work (tests/test_gpu_parity.py: bit-equal)
        "ORBX_GRAPH",          # 0: plain launches instead of the captured hipGraph
        "ORBX_SELECT_SPREAD",  # fused vs three-kernel selection
        "ORBX_BLUR_IMPL",      # which separable blur kernel: LDS tiles / 4 pixels per lane (default) / 16 pixels per lane
        "ORBX_FUSE",           # 0: pyramid and blur as two kernels instead of one
        "ORBX_PYR_GROUP",      # frames per dispatch group of the fused kernel (tests/test_batch64_parity.py)
        "ORBX_FAST_CHUNK",     # tiles per FAST workgroup (tests/test_batch64_parity.py)
        "ORBX_FAST_IMPL",      # 3: LDS tile FAST kernel instead of the register-streaming one (both parity modules run both)
        "ORBX_TOP_ROWS",       # FAST tile rows of the first pass of the top-rows-first pipeline, 0: one pass (same test)
    }
    assert found <= allowed, sorted(found - allowed)
