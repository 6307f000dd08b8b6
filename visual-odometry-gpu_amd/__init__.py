"""MI355X-native ORB feature front-end (drop-in for the reference's orb.hpp path).

The product is liborbx.so (HIP kernels + C ABI, csrc/); `orbx` is its ctypes view.
The directory name contains a hyphen, so import it with
    importlib.import_module("visual-odometry-gpu_amd")
(see __graft_entry__.load_package()).
"""
from . import orbx, shard, streams  # noqa: F401
from .orbx import Context, OrbxError, default_params  # noqa: F401
