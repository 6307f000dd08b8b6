import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The native artefacts are git-ignored; build them if this checkout has none yet."""
    need = [os.path.join(ROOT, "visual-odometry-gpu_amd", "liborbx.so"), os.path.join(ROOT, "oracle", "liborb_oracle.so"),
            os.path.join(ROOT, "tests", "cpp", "test_host_mirror.bin")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def pkg():
    _ensure_built()
    return importlib.import_module("visual-odometry-gpu_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib
