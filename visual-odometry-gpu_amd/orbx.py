"""ctypes binding of liborbx.so (include/orbx.h) for tests and bench.py.

The product is the C-ABI library; this module is only the thinnest possible
Python view of it (numpy arrays in / out, device pointers as ints).  It never
falls back to a CPU implementation: if liborbx.so is missing or no gfx950
device is present, loading / Context creation raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(_HERE, "liborbx.so")  # ORBX_LIB: A/B builds (diagnostics)

MAX_LEVELS = 16
OK, ERR_INVALID_ARG, ERR_CAPACITY, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED = range(6)
SELECT_HARRIS, SELECT_ROWMAJOR = 0, 1
BLUR_NONE, BLUR_UPPER, BLUR_ALL = 0, 1, 2
BLUR_SEP16, BLUR_K273 = 0, 1
STAGE_PYRAMID, STAGE_BLUR, STAGE_FAST, STAGE_COMPACT, STAGE_HARRIS, STAGE_SELECT, STAGE_DESCRIBE = range(7)
NUM_STAGE_TIMES = 8
STAGE_NAMES = ["pyramid", "blur", "fast_nms", "compact", "harris", "select", "describe", "total"]

# every symbol include/orbx.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "orbx_params_default_gpu", "orbx_params_default_cpu", "orbx_create", "orbx_destroy",
    "orbx_last_error_string", "orbx_status_string", "orbx_version", "orbx_get_plan",
    "orbx_detect_and_compute", "orbx_detect_and_compute_batch_device", "orbx_detect_and_compute_batch_host",
    "orbx_wait", "orbx_batch_results_device", "orbx_batch_results_host", "orbx_batch_fetch", "orbx_batch_prefetch", "orbx_batch_prefetch_compact",
    "orbx_batch_fetch_previous", "orbx_enable_stage_timing",
    "orbx_last_stage_times", "orbx_stage_times_history", "orbx_bench_stage", "orbx_set_fast_early_exit",
    "orbx_set_fused_pyramid_blur", "orbx_set_top_rows_first", "orbx_set_pipelined_batches", "orbx_set_host_results", "orbx_fast_tile_counts", "orbx_pyramid_pixel_counts", "orbx_lk_track", "orbx_lk_pyramid_levels", "orbx_fast_score", "orbx_nms", "orbx_fast",
    "orbx_orientations", "orbx_brief", "orbx_harris", "orbx_blur5_sep", "orbx_blur5_273", "orbx_conv2d",
    "orbx_gaussian_blur_conv", "orbx_gaussian_kernel", "orbx_sobel", "orbx_build_pyramid_level",
    "orbx_select_top", "orbx_knn2", "orbx_match_ratio", "orbx_batch_match_consecutive", "orbx_batch_match_fetch",
]


class Params(C.Structure):
    _fields_ = [
        ("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
        ("threshold", C.c_int32), ("n", C.c_int32), ("nms_window", C.c_int32), ("patch_size", C.c_int32),
        ("harris_window", C.c_int32), ("harris_k", C.c_float), ("select_mode", C.c_int32),
        ("blur_levels", C.c_int32), ("blur_kind", C.c_int32), ("max_width", C.c_int32),
        ("max_height", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
    ]


class BatchView(C.Structure):
    _fields_ = [
        ("counts", C.c_void_p), ("keypoints", C.c_void_p), ("level_kps", C.c_void_p),
        ("orientations", C.c_void_p), ("responses", C.c_void_p), ("levels", C.c_void_p),
        ("descriptors", C.c_void_p), ("slot_capacity", C.c_int32), ("n", C.c_int32),
        ("keypoints16", C.c_void_p),
    ]


class OrbxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("liborbx status %d (%s): %s" % (status, _status_string(status), msg))
        self.status = status


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP
    runtimes in one process cannot both open the GPU, so when torch is installed
    pre-load ITS runtime (same soname, libamdhip64.so.7) before liborbx.so; the
    dynamic loader then binds liborbx's NEEDED entry to it.  Without torch, or
    with ORBX_HIP_RUNTIME=system, liborbx uses /opt/rocm's runtime."""
    if os.environ.get("ORBX_HIP_RUNTIME", "torch") != "torch":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load liborbx.so (raises OSError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("liborbx.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C visual-odometry-gpu_amd/csrc` (there is no CPU fallback)")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(LIB_PATH)
        lib.orbx_last_error_string.restype = C.c_char_p
        lib.orbx_last_error_string.argtypes = [C.c_void_p]
        lib.orbx_status_string.restype = C.c_char_p
        lib.orbx_version.restype = C.c_char_p
        lib.orbx_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        lib.orbx_destroy.argtypes = [C.c_void_p]
        lib.orbx_destroy.restype = None
        _lib = lib
    return _lib


def _status_string(st):
    try:
        return load().orbx_status_string(int(st)).decode()
    except OSError:
        return "?"


def default_params(flavour="gpu", **kw):
    """orbx_params with the reference defaults of the GPU (orb.hpp) or CPU (orb_cpu.hpp) flavour."""
    p = Params()
    fn = load().orbx_params_default_gpu if flavour == "gpu" else load().orbx_params_default_cpu
    fn(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("image must be 2-D uint8")
    return a


def _kps(k):
    k = np.ascontiguousarray(k, dtype=np.int32).reshape(-1, 2)
    return k


class Context:
    """One orbx_ctx: single-threaded, owns all device memory, one per GPU."""

    def __init__(self, params=None, **kw):
        self._lib = load()
        if params is None:
            params = default_params(kw.pop("flavour", "gpu"), **kw)
        self.params = params
        h = C.c_void_p()
        st = self._lib.orbx_create(C.byref(params), C.byref(h))
        if st != OK:
            raise OrbxError(st, self._lib.orbx_last_error_string(None).decode())
        self._h = h
        self._cap_cache = {}
        self._dac = self._lib.orbx_detect_and_compute
        self._dac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [
            C.c_int, C.POINTER(C.c_int)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.orbx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st, allow=()):
        if st != OK and st not in allow:
            raise OrbxError(st, self._lib.orbx_last_error_string(self._h).decode())
        return st

    # ---- geometry
    def plan(self, width, height):
        nl = self.params.nlevels
        lw = np.zeros(nl, np.int32)
        lh = np.zeros(nl, np.int32)
        q = np.zeros(nl, np.int32)
        cap = np.zeros(nl, np.int32)
        sc = np.zeros(nl, np.float32)
        oc = C.c_int32(0)
        self._chk(self._lib.orbx_get_plan(self._h, width, height, _ptr(lw), _ptr(lh), _ptr(q), _ptr(cap), _ptr(sc),
                                          C.byref(oc)))
        return dict(level_w=lw, level_h=lh, quota=q, fast_cap=cap, scale=sc, out_capacity=oc.value)

    # ---- whole path
    def detect_and_compute(self, image, capacity=None):
        """ORB::detectAndCompute (orb.hpp:37). Returns dict of numpy arrays."""
        image = _img(image)
        h, w = image.shape
        if capacity is None:
            capacity = self._cap_cache.get((w, h))
            if capacity is None:
                capacity = self._cap_cache[(w, h)] = max(self.plan(w, h)["out_capacity"], 1)
        kps = np.empty((capacity, 2), np.int32)
        lkp = np.empty((capacity, 2), np.int32)
        ang = np.empty(capacity, np.float32)
        resp = np.empty(capacity, np.float32)
        lev = np.empty(capacity, np.int32)
        desc = np.empty((capacity, 32), np.uint8)
        cnt = C.c_int(0)
        st = self._dac(self._h, image.ctypes.data, w, h, image.strides[0], kps.ctypes.data, ang.ctypes.data,
                       desc.ctypes.data, resp.ctypes.data, lev.ctypes.data, lkp.ctypes.data, capacity, C.byref(cnt))
        if st != OK and st != ERR_CAPACITY:
            self._chk(st)
        c = min(cnt.value, capacity)
        return dict(count=cnt.value, status=st, kps=kps[:c], kps_level=lkp[:c], angles=ang[:c], responses=resp[:c],
                    levels=lev[:c], desc=desc[:c])

    def batch_device(self, d_ptr, n, width, height, row_stride=None, frame_stride=None, stream=None):
        row_stride = width if row_stride is None else row_stride
        frame_stride = row_stride * height if frame_stride is None else frame_stride
        self._chk(self._lib.orbx_detect_and_compute_batch_device(
            self._h, C.c_void_p(d_ptr), n, width, height, row_stride, C.c_size_t(frame_stride),
            C.c_void_p(stream) if stream else None))

    def batch_host(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim != 3:
            raise ValueError("frames must be (n, h, w) uint8")
        n, h, w = frames.shape
        self._chk(self._lib.orbx_detect_and_compute_batch_host(self._h, _ptr(frames), n, w, h, w, C.c_size_t(w * h)))

    def wait(self):
        self._chk(self._lib.orbx_wait(self._h))

    def batch_view(self):
        v = BatchView()
        self._chk(self._lib.orbx_batch_results_device(self._h, C.byref(v)))
        return v

    def batch_host_view(self, previous=False):
        """Zero-copy numpy views of the pinned host mirror of a result block (orbx_batch_results_host); previous: how
        many batches back (False / 0: the last one, True / 1: the one before, up to 3)."""
        v = BatchView()
        self._chk(self._lib.orbx_batch_results_host(self._h, int(previous), C.byref(v)))
        n, cap = v.n, v.slot_capacity

        def arr(ptr, dtype, shape):
            if not ptr:  # (a section a compact prefetch did not copy)
                return None
            size = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * size).from_address(ptr), dtype=dtype).reshape(shape)

        # kps16: the level-0 coordinates as (x, y) uint16 pairs (x | y << 16, little endian); the only keypoint
        # section of a compact copy
        return dict(counts=arr(v.counts, np.int32, (n,)), kps=arr(v.keypoints, np.int32, (n, cap, 2)),
                    kps16=arr(v.keypoints16, np.uint16, (n, cap, 2)),
                    kps_level=arr(v.level_kps, np.int32, (n, cap, 2)), angles=arr(v.orientations, np.float32, (n, cap)),
                    responses=arr(v.responses, np.float32, (n, cap)), levels=arr(v.levels, np.int32, (n, cap)),
                    desc=arr(v.descriptors, np.uint8, (n, cap, 32)))

    def batch_prefetch(self, compact=False):
        """Start the asynchronous D2H copy of the last batch's result block (overlaps the next batch); compact: only
        counts, packed keypoints (kps16), orientations and descriptors (orbx_batch_prefetch_compact)."""
        if compact:
            self._chk(self._lib.orbx_batch_prefetch_compact(self._h))
        else:
            self._chk(self._lib.orbx_batch_prefetch(self._h))

    def batch_fetch(self, first, n, capacity, previous=False):
        """Results of the last batch (previous=True: of the batch before it, see orbx_batch_fetch_previous)."""
        counts = np.zeros(n, np.int32)
        kps = np.zeros((n, capacity, 2), np.int32)
        lkp = np.zeros((n, capacity, 2), np.int32)
        ang = np.zeros((n, capacity), np.float32)
        resp = np.zeros((n, capacity), np.float32)
        lev = np.zeros((n, capacity), np.int32)
        desc = np.zeros((n, capacity, 32), np.uint8)
        fn = self._lib.orbx_batch_fetch_previous if previous else self._lib.orbx_batch_fetch
        st = fn(self._h, first, n, _ptr(counts), _ptr(kps), _ptr(ang), _ptr(desc), _ptr(resp), _ptr(lev), _ptr(lkp),
                capacity)
        self._chk(st, allow=(ERR_CAPACITY,))
        return dict(counts=counts, kps=kps, kps_level=lkp, angles=ang, responses=resp, levels=lev, desc=desc,
                    status=st)

    def enable_stage_timing(self, mode=1):
        """0/False off, 1/True events around every stage, 2 only around blur and fast+nms."""
        self._chk(self._lib.orbx_enable_stage_timing(self._h, int(mode)))

    def set_fast_early_exit(self, on=True):
        self._chk(self._lib.orbx_set_fast_early_exit(self._h, 1 if on else 0))

    def set_fused_pyramid_blur(self, on=True):
        self._chk(self._lib.orbx_set_fused_pyramid_blur(self._h, 1 if on else 0))

    def set_host_results(self, on=True):
        """The describe kernel also writes the compact record into the pinned host mirror (orbx_set_host_results)."""
        self._chk(self._lib.orbx_set_host_results(self._h, 1 if on else 0))

    def set_pipelined_batches(self, on=True):
        """Consecutive batch_device calls alternate between two lanes (own stream, own pools) and overlap."""
        self._chk(self._lib.orbx_set_pipelined_batches(self._h, 1 if on else 0))

    def set_top_rows_first(self, mode=2):
        """0: one pass, 1: the pyramid top rows first whenever eligible, 2: adaptive (default)."""
        self._chk(self._lib.orbx_set_top_rows_first(self._h, int(mode)))

    def lk_track(self, prev, nxt, pts, win=21, max_level=3, max_iters=30, epsilon=0.01):
        """cv::calcOpticalFlowPyrLK(prev, next, pts, ...) as called at feature_tracking.cpp:175-181.
        prev=None: the previous call's `next` image is this call's `prev`.  Returns next_pts, status, err."""
        nxt = _img(nxt)
        h, w = nxt.shape
        if prev is not None:
            prev = _img(prev)
            if prev.shape != nxt.shape:
                raise ValueError("prev and next must have the same size")
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = len(pts)
        out = np.zeros((max(n, 1), 2), np.float32)
        st = np.zeros(max(n, 1), np.uint8)
        err = np.zeros(max(n, 1), np.float32)
        f = self._lib.orbx_lk_track
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
        self._chk(f(self._h, _ptr(prev), w, _ptr(nxt), w, w, h, _ptr(pts), n, _ptr(out), _ptr(st), _ptr(err), win,
                    max_level, max_iters, epsilon))
        return out[:n].copy(), st[:n].copy(), err[:n].copy()

    def fast_tile_counts(self):
        """(tiles that did the full FAST/NMS work, all tiles) of the last whole-path batch."""
        w, t = C.c_longlong(0), C.c_longlong(0)
        self._chk(self._lib.orbx_fast_tile_counts(self._h, C.byref(w), C.byref(t)))
        return w.value, t.value

    def pyramid_pixel_counts(self):
        """(pyramid pixels produced, all pyramid pixels) of the last whole-path batch (top-rows-first pipeline)."""
        w, t = C.c_longlong(0), C.c_longlong(0)
        self._chk(self._lib.orbx_pyramid_pixel_counts(self._h, C.byref(w), C.byref(t)))
        return w.value, t.value

    def last_stage_times(self, back=0):
        """Stage times (ms) of the timed batched call `back` calls ago; call wait() first."""
        ms = np.zeros(NUM_STAGE_TIMES, np.float32)
        self._chk(self._lib.orbx_stage_times_history(self._h, back, _ptr(ms)))
        return dict(zip(STAGE_NAMES, ms.tolist()))

    def bench_stage(self, n_frames, stage, reps):
        ms = C.c_float(0)
        self._chk(self._lib.orbx_bench_stage(self._h, n_frames, stage, reps, C.byref(ms)))
        return ms.value

    # ---- stage operators (names follow the reference's free functions)
    def fast_score(self, image, threshold, n=9):
        image = _img(image)
        h, w = image.shape
        out = np.zeros((h, w), np.float32)
        self._chk(self._lib.orbx_fast_score(self._h, _ptr(image), w, h, image.strides[0], threshold, n, _ptr(out)))
        return out

    def nms(self, scores, nms_window, nfeatures, threshold=0.0):
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        h, w = scores.shape
        kps = np.zeros((max(nfeatures, 1), 2), np.int32)
        cnt, tot = C.c_int(0), C.c_int(0)
        self._chk(self._lib.orbx_nms(self._h, _ptr(scores), w, h, nms_window, nfeatures, C.c_float(threshold),
                                     _ptr(kps), C.byref(cnt), C.byref(tot)))
        return kps[:cnt.value].copy(), tot.value

    def fast(self, image, threshold, n, nms_window, nfeatures):
        image = _img(image)
        h, w = image.shape
        kps = np.zeros((max(nfeatures, 1), 2), np.int32)
        cnt, tot = C.c_int(0), C.c_int(0)
        self._chk(self._lib.orbx_fast(self._h, _ptr(image), w, h, image.strides[0], threshold, n, nms_window,
                                      nfeatures, _ptr(kps), C.byref(cnt), C.byref(tot)))
        return kps[:cnt.value].copy(), tot.value

    def orientations(self, image, kps, patch_size):
        image = _img(image)
        h, w = image.shape
        kps = _kps(kps)
        out = np.zeros(len(kps), np.float32)
        self._chk(self._lib.orbx_orientations(self._h, _ptr(image), w, h, image.strides[0], _ptr(kps), len(kps),
                                              patch_size, _ptr(out)))
        return out

    def brief(self, image, kps, angles):
        image = _img(image)
        h, w = image.shape
        kps = _kps(kps)
        angles = np.ascontiguousarray(angles, dtype=np.float32)
        out = np.zeros((len(kps), 32), np.uint8)
        self._chk(self._lib.orbx_brief(self._h, _ptr(image), w, h, image.strides[0], _ptr(kps), _ptr(angles), len(kps),
                                       _ptr(out)))
        return out

    def harris(self, image, kps, window=7, k=0.04):
        image = _img(image)
        h, w = image.shape
        kps = _kps(kps)
        out = np.zeros(len(kps), np.float32)
        self._chk(self._lib.orbx_harris(self._h, _ptr(image), w, h, image.strides[0], _ptr(kps), len(kps), window,
                                        C.c_float(k), _ptr(out)))
        return out

    def blur5_sep(self, image):
        image = _img(image)
        h, w = image.shape
        out = np.zeros((h, w), np.uint8)
        self._chk(self._lib.orbx_blur5_sep(self._h, _ptr(image), w, h, image.strides[0], _ptr(out), w))
        return out

    def blur5_273(self, image):
        image = _img(image)
        h, w = image.shape
        out = np.zeros((h, w), np.uint8)
        self._chk(self._lib.orbx_blur5_273(self._h, _ptr(image), w, h, image.strides[0], _ptr(out), w))
        return out

    def conv2d(self, padded, kernel):
        padded = _img(padded)
        h, w = padded.shape
        kernel = np.ascontiguousarray(kernel, dtype=np.float32)
        K = kernel.shape[0]
        out = np.zeros((h - K + 1, w - K + 1), np.uint8)
        self._chk(self._lib.orbx_conv2d(self._h, _ptr(padded), w, h, padded.strides[0], _ptr(kernel), K, _ptr(out)))
        return out

    def gaussian_blur_conv(self, image, K):
        image = _img(image)
        h, w = image.shape
        out = np.zeros((h, w), np.uint8)
        self._chk(self._lib.orbx_gaussian_blur_conv(self._h, _ptr(image), w, h, image.strides[0], K, _ptr(out)))
        return out

    def sobel(self, image, direction):
        image = _img(image)
        h, w = image.shape
        out = np.zeros((h, w), np.uint8)
        self._chk(self._lib.orbx_sobel(self._h, _ptr(image), w, h, image.strides[0], direction, _ptr(out)))
        return out

    def build_pyramid_level(self, image, level):
        image = _img(image)
        h, w = image.shape
        pl = self.plan(w, h)
        out = np.zeros((int(pl["level_h"][level]), int(pl["level_w"][level])), np.uint8)
        lw, lh = C.c_int(0), C.c_int(0)
        self._chk(self._lib.orbx_build_pyramid_level(self._h, _ptr(image), w, h, image.strides[0], level, _ptr(out),
                                                     C.byref(lw), C.byref(lh)))
        assert (lh.value, lw.value) == out.shape
        return out

    def select_top(self, responses, keep):
        responses = np.ascontiguousarray(responses, dtype=np.float32)
        idx = np.zeros(max(len(responses), 1), np.int32)
        kept = C.c_int(0)
        self._chk(self._lib.orbx_select_top(self._h, _ptr(responses), len(responses), keep, _ptr(idx), C.byref(kept)))
        return idx[:kept.value].copy()


def _desc(d):
    d = np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, 32)
    return d


def _matcher_methods():
    def knn2(self, query, train):
        """flann->knnMatch(des1, des2, matches, 2) as exact Hamming 2-NN: (idx[nq,2], dist[nq,2])."""
        q, t = _desc(query), _desc(train)
        idx = np.full((max(len(q), 1), 2), -1, np.int32)
        dist = np.full((max(len(q), 1), 2), -1, np.int32)
        self._chk(self._lib.orbx_knn2(self._h, _ptr(q), len(q), _ptr(t), len(t), _ptr(idx), _ptr(dist)))
        return idx[:len(q)].copy(), dist[:len(q)].copy()

    def match_ratio(self, query, train, ratio=0.8):
        """knnMatch + `m.distance < ratio * n.distance` (feature_matching.cpp:166-181)."""
        q, t = _desc(query), _desc(train)
        n = max(len(q), 1)
        qi, ti, d1 = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        cnt = C.c_int(0)
        self._chk(self._lib.orbx_match_ratio(self._h, _ptr(q), len(q), _ptr(t), len(t), C.c_double(ratio), _ptr(qi),
                                             _ptr(ti), _ptr(d1), n, C.byref(cnt)))
        m = cnt.value
        return qi[:m].copy(), ti[:m].copy(), d1[:m].copy()

    def batch_match_consecutive(self, ratio=0.8):
        self._chk(self._lib.orbx_batch_match_consecutive(self._h, C.c_double(ratio)))

    def batch_match_fetch(self, pair, capacity):
        qi, ti, d1 = np.zeros(capacity, np.int32), np.zeros(capacity, np.int32), np.zeros(capacity, np.int32)
        cnt = C.c_int(0)
        self._chk(self._lib.orbx_batch_match_fetch(self._h, pair, _ptr(qi), _ptr(ti), _ptr(d1), capacity,
                                                   C.byref(cnt)))
        m = cnt.value
        return qi[:m].copy(), ti[:m].copy(), d1[:m].copy()

    for f in (knn2, match_ratio, batch_match_consecutive, batch_match_fetch):
        setattr(Context, f.__name__, f)


_matcher_methods()


def gaussian_kernel(K, sigma=-1.0):
    out = np.zeros(K * K, np.float32)
    st = load().orbx_gaussian_kernel(K, C.c_float(sigma), _ptr(out))
    if st != OK:
        raise OrbxError(st, "orbx_gaussian_kernel")
    return out.reshape(K, K)


def version():
    return load().orbx_version().decode()
