"""ctypes binding of libslamhip.so (include/slam_hip.h).

This is the binding a maintainer of the reference would add next to
course_agv_slam/scripts: the reference has no FFI layer of its own, so
the C ABI is bound here one function per reference method.

The library is the ONLY implementation of the hot path: there is no CPU fallback.  If the
shared object is missing, or no gfx950 device is visible when a context is created, the
package raises instead of computing anything on the host.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAM_HIP_LIB") or os.path.join(_HERE, "libslamhip.so")   # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "slam_hip.h")

SLAM_OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_NAN, ERR_OVERFLOW, ERR_NODEVICE = 0, -1, -2, -3, -4, -5, -6
F64, F32, F16 = 0, 1, 2
DTYPES = {"f64": F64, "f32": F32, "f16": F16, np.float64: F64, np.float32: F32, np.float16: F16}
NP_DTYPES = {F64: np.float64, F32: np.float32, F16: np.float16}
K_NAMES = ("points", "icp", "compose", "grid", "finalize", "nn", "kabsch", "bresenham")


class SlamError(RuntimeError):
    """A libslamhip call failed (bad argument, HIP error, no device)."""


class LibraryMissing(ImportError):
    """libslamhip.so has not been built; there is no fallback implementation."""


_lib = None
_lib_lock = threading.Lock()

_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_SIGS = {
    "slam_abi_version": ([], _i),
    "slam_last_error": ([], C.c_char_p),
    "slam_create": ([_i, _vp, C.POINTER(_vp)], _i),
    "slam_destroy": ([_vp], _i),
    "slam_synchronize": ([_vp], _i),
    "slam_stream_order": ([_vp, _vp, _i], _i),
    "slam_check_status": ([_vp], _i),
    "slam_set_option": ([_vp, C.c_char_p, _d], _i),
    "slam_timing_enable": ([_vp, _i], _i),
    "slam_timing_read": ([_vp, _vp, _vp], _i),
    "slam_scan_to_points": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "slam_scan_to_points_dev": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "slam_nn": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp], _i),
    "slam_nn_dev": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp], _i),
    "slam_kabsch2d": ([_vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_kabsch2d_dev": ([_vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_icp_batch": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _d, _vp, _vp, _vp], _i),
    "slam_icp_batch_dev": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _d, _vp, _vp, _vp], _i),
    "slam_pose_compose": ([_vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_pose_compose_dev": ([_vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_grid_create": ([_vp, _i, _i, _i, _d, _d, _d, _d, _d, _d, C.POINTER(_vp)], _i),
    "slam_grid_destroy": ([_vp, _vp], _i),
    "slam_grid_reset": ([_vp, _vp], _i),
    "slam_grid_update": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_grid_update_dev": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_grid_read": ([_vp, _vp, _i, _vp, _vp, _vp, _vp], _i),
    "slam_grid_finalize_dev": ([_vp, _vp, _vp], _i),
    "slam_grid_occupancy_data": ([_vp, _vp, _i, _vp], _i),
    "slam_grid_visits": ([_vp, _vp, C.POINTER(C.c_uint64)], _i),
    "slam_bresenham_batch": ([_vp, _vp, _vp, _i, _vp, _vp, _vp, C.c_int64], _i),
    "slam_replay": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "slam_particles": ([_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp], _i),
    "slam_particles_dev": ([_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp], _i),
    "slam_grid_counters_dev": ([_vp, _vp, C.POINTER(_vp), C.POINTER(_vp)], _i),
    "slam_grid_live_pmap": ([_vp, _vp, C.POINTER(_vp)], _i),
    "slam_grid_update_scans": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i], _i),
    "slam_grid_update_scans_dev": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i], _i),
    "slam_map_obstacles": ([_vp, _vp, _i, _i, _i, _d, _d, _d, _vp, _vp, _i, C.POINTER(_i)], _i),
    "slam_map_obstacles_dev": ([_vp, _vp, _i, _i, _i, _d, _d, _d, _vp, _vp, _i, _vp], _i),
    "slam_virtual_scan": ([_vp, _vp, _vp, _i, _vp, _i, _d, _d, _i, _vp], _i),
    "slam_virtual_scan_dev": ([_vp, _vp, _vp, _i, _vp, _i, _d, _d, _i, _vp], _i),
    "slam_scan_to_points_f64": ([_vp, _vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_scan_to_points_f64_dev": ([_vp, _vp, _vp, _vp, _i, _i, _vp], _i),
    "slam_map_observation": ([_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _d, _d, _i, _d, _vp, _vp], _i),
    "slam_map_observation_dev": ([_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _d, _d, _i, _d, _vp, _vp, _vp, _vp], _i),
    "slam_replay_dev": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
}


def header_symbols(path=HEADER_PATH):
    """Names of every function include/slam_hip.h declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load libslamhip.so once.  Raises LibraryMissing if it was never built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise LibraryMissing(
                    "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C <package>/csrc`). There is no CPU fallback for this path." % LIB_PATH)
            # One HIP runtime per process: PyTorch-ROCm wheels bundle their own
            # libamdhip64.so.7.  If torch is importable it is loaded FIRST, so that this
            # library's libamdhip64.so.7 dependency resolves to the copy torch uses and
            # device pointers / streams can be shared (bench.py, torch.distributed).  Loading
            # them in the other order leaves torch without a usable device.
            if os.environ.get("SLAM_HIP_STANDALONE") != "1":
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
            L = C.CDLL(LIB_PATH)
            for name, (args, res) in _SIGS.items():
                fn = getattr(L, name)
                fn.argtypes, fn.restype = args, res
            if L.slam_abi_version() != 1:
                raise SlamError("libslamhip ABI version %d, binding expects 1" % L.slam_abi_version())
            _lib = L
    return _lib


def _raise(code):
    msg = (lib().slam_last_error() or b"").decode("utf-8", "replace")
    if code == ERR_NAN:
        raise ValueError(msg)            # int(nan) in the reference (mapping.py:33)
    if code == ERR_OVERFLOW:
        raise OverflowError(msg)         # int(inf) in the reference (mapping.py:33)
    if code == ERR_NOMEM:
        raise MemoryError(msg)
    raise SlamError("libslamhip error %d: %s" % (code, msg))


def check(code):
    if code != SLAM_OK:
        _raise(code)


def ptr(a):
    """Host pointer of a C-contiguous ndarray, device pointer of a torch tensor, or an
    int passed through; None -> NULL."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return int(a)


class Context:
    """Owns a slam_ctx (one HIP stream + device workspace).  Not thread-safe: one per
    host thread, like the reference's single rospy callback thread."""

    def __init__(self, device=0, stream=None):
        h = _vp()
        check(lib().slam_create(int(device), _vp(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def handle(self):
        if self._h is None:
            raise SlamError("context already destroyed")
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None:
            lib().slam_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().slam_synchronize(self.handle))

    def check_status(self):
        check(lib().slam_check_status(self.handle))

    def stream_order(self, stream, direction):
        """Order this context's work against another HIP stream (``stream``: a raw hipStream_t handle, e.g.
        ``torch.cuda.current_stream().cuda_stream``) without a host synchronise - ``slam_stream_order``:
        direction 0 = that stream waits for everything enqueued here so far, 1 = work enqueued here from now on
        waits for what that stream holds now."""
        check(lib().slam_stream_order(self.handle, _vp(stream) if stream else None, int(direction)))

    def set_option(self, name, value):
        check(lib().slam_set_option(self.handle, name.encode(), float(value)))

    def timing_enable(self, on=True, only=None):
        """``only``: family names (K_NAMES) whose launches carry events; None: every family."""
        code = int(bool(on))
        if on and only is not None:
            code = 2 * sum(1 << K_NAMES.index(k) for k in only)
        check(lib().slam_timing_enable(self.handle, code))

    def timing_read(self):
        """{family: (milliseconds, launches)} since the last read."""
        ms = np.zeros(len(K_NAMES), dtype=np.float64)
        cnt = np.zeros(len(K_NAMES), dtype=np.int64)
        check(lib().slam_timing_read(self.handle, ptr(ms), ptr(cnt)))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(K_NAMES)}


_default = {}
_default_lock = threading.Lock()


def default_context(device=0):
    """Process-wide context used by the drop-in classes when none is given."""
    with _default_lock:
        c = _default.get(device)
        if c is None or c._h is None:
            c = _default[device] = Context(device)
        return c


def trig_tables(angle_min, angle_max, n):
    """cos/sin of the beam angles exactly as the reference forms them
    (icp.py:227-228): numpy.linspace then numpy.cos / numpy.sin."""
    ang = np.linspace(angle_min, angle_max, n)
    return np.ascontiguousarray(np.cos(ang)), np.ascontiguousarray(np.sin(ang))
