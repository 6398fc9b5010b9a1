"""ctypes access to oracle/liboracle.so (the C restatement, oracle/slam_oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_bp = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")


def build(force=False):
    """Build (if stale) and return the path of the shared object; SLAM_ORACLE_LIB selects another
    target of oracle/Makefile (liboracle_san.so: the AddressSanitizer / UBSan build)."""
    name = os.environ.get("SLAM_ORACLE_LIB", "liboracle.so")
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "slam_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, name])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_laser_to_points.argtypes = [_fp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.orc_find_nearest.argtypes = [_dp, _dp, C.c_int, _dp, _dp, C.c_int, _dp, _ip]
        L.orc_set_nn_rule.argtypes = [C.c_int]
        L.orc_set_nn_rule.restype = None
        L.orc_nn_rule_splits.argtypes = [C.c_int]
        L.orc_nn_rule_splits.restype = C.c_long
        L.orc_get_transform.argtypes = [_dp, _dp, _dp, _dp, C.c_int, _dp]
        L.orc_icp_process.argtypes = [_dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp, C.POINTER(C.c_double)]
        L.orc_icp_process.restype = C.c_int
        L.orc_icp_batch.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _ip, _dp]
        L.orc_compose_pose.argtypes = [_dp, _dp]
        L.orc_world_points.argtypes = [_dp, _dp, _dp, C.c_int, _dp, _dp]
        L.orc_bresenham.argtypes = [C.c_int] * 4 + [_ip, C.c_int]
        L.orc_walk_end_mismatches.argtypes = [C.c_int]
        L.orc_walk_end_mismatches.restype = C.c_long
        L.orc_bresenham.restype = C.c_int
        L.orc_grid_create.argtypes = [C.c_int, C.c_int] + [C.c_double] * 6
        L.orc_grid_create.restype = C.c_void_p
        L.orc_grid_destroy.argtypes = [C.c_void_p]
        for f, t in (("datamap", C.c_double), ("pmap", C.c_int8), ("pass", C.c_uint32), ("hit", C.c_uint32)):
            fn = getattr(L, "orc_grid_" + f)
            fn.argtypes = [C.c_void_p]
            fn.restype = C.POINTER(t)
        L.orc_grid_update.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_double, C.c_double]
        L.orc_grid_update.restype = C.c_long
        L.orc_pass_count_threshold.argtypes = [C.c_double, C.c_double]
        L.orc_pass_count_threshold.restype = C.c_int
        L.orc_occupancy_grid_data.argtypes = [C.c_void_p, _bp]
        L.orc_replay.argtypes = [_fp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double, _dp, C.c_void_p, _dp, _dp, _ip, C.c_int]
        L.orc_replay.restype = C.c_long
        L.orc_laser_estimation.argtypes = [_dp, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_map_obstacles.argtypes = [_bp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.orc_map_obstacles.restype = C.c_int
        L.orc_replay_mt.argtypes = L.orc_replay.argtypes
        L.orc_replay_mt.restype = C.c_long
        _LIB = L
    return _LIB


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def trig_tables(angle_min, angle_max, n):
    ang = np.linspace(angle_min, angle_max, n)
    return np.cos(ang), np.sin(ang)


def laser_to_points(ranges, angle_min, angle_max, clip_inf=True):
    r = _c(ranges, np.float32)
    ct, st = trig_tables(angle_min, angle_max, r.shape[0])
    x, y = np.empty(r.shape[0]), np.empty(r.shape[0])
    lib().orc_laser_to_points(r, _c(ct), _c(st), r.shape[0], int(clip_inf), x, y)
    return x, y


def find_nearest(src, tar):
    """src [N,2], tar [M,2] -> (distances, indices)."""
    sx, sy, tx, ty = _c(src[:, 0]), _c(src[:, 1]), _c(tar[:, 0]), _c(tar[:, 1])
    d, i = np.empty(sx.shape[0]), np.empty(sx.shape[0], dtype=np.int32)
    lib().orc_find_nearest(sx, sy, sx.shape[0], tx, ty, tx.shape[0], d, i)
    return d, i


def get_transform(src, tar):
    T = np.empty(9)
    lib().orc_get_transform(_c(src[:, 0]), _c(src[:, 1]), _c(tar[:, 0]), _c(tar[:, 1]), src.shape[0], T)
    return T.reshape(3, 3)


def icp_process(tar_pc, src_pc, max_iter=30, tolerance=0.001):
    """tar_pc, src_pc [3,N] or [2,N] -> (T, iters, mean_err)."""
    T = np.empty(9)
    me = C.c_double(0.0)
    it = lib().orc_icp_process(_c(tar_pc[0]), _c(tar_pc[1]), tar_pc.shape[1], _c(src_pc[0]), _c(src_pc[1]),
                               src_pc.shape[1], max_iter, tolerance, T, C.byref(me))
    return T.reshape(3, 3), it, me.value


def icp_batch(tar, src, max_iter=30, tolerance=0.001):
    """tar [B,2,M], src [B,2,N] -> (T [B,3,3], iters [B], mean_err [B]); OpenMP over B."""
    tar, src = _c(tar), _c(src)
    B = tar.shape[0]
    T, it, me = np.empty((B, 9)), np.empty(B, dtype=np.int32), np.empty(B)
    lib().orc_icp_batch(tar, src, B, tar.shape[2], src.shape[2], max_iter, tolerance, T, it, me)
    return T.reshape(B, 3, 3), it, me


def compose_pose(sta, T):
    s = _c(sta).copy()
    lib().orc_compose_pose(s, _c(T).reshape(-1))
    return s


def world_points(pose, px, py):
    ox, oy = np.empty(len(px)), np.empty(len(px))
    lib().orc_world_points(_c(pose), _c(px), _c(py), len(px), ox, oy)
    return ox, oy


def bresenham(start, end):
    cap = max(abs(int(end[0]) - int(start[0])), abs(int(end[1]) - int(start[1]))) + 1
    xy = np.empty(2 * cap, dtype=np.int32)
    n = lib().orc_bresenham(int(start[0]), int(start[1]), int(end[0]), int(end[1]), xy, cap)
    return xy[: 2 * n].reshape(-1, 2)


def walk_end_mismatches(max_dx):
    """(dx, dy) pairs up to max_dx whose float-error walk does not end in the cell of its other end."""
    return int(lib().orc_walk_end_mismatches(int(max_dx)))


class Grid:
    def __init__(self, xw, yw, scale=10.0, off_x=10.0, off_y=10.0, free_inc=0.01, hit_inc=20.0, thresh=10.0):
        self.xw, self.yw = xw, yw
        self._g = lib().orc_grid_create(xw, yw, scale, off_x, off_y, free_inc, hit_inc, thresh)
        self.visits = 0

    def __del__(self):
        if getattr(self, "_g", None):
            lib().orc_grid_destroy(self._g)
            self._g = None

    def _view(self, name, dt):
        p = getattr(lib(), "orc_grid_" + name)(self._g)
        return np.ctypeslib.as_array(p, shape=(self.xw, self.yw)).view(dt)

    pmap = property(lambda self: self._view("pmap", np.int8))
    datamap = property(lambda self: self._view("datamap", np.float64))
    pass_cnt = property(lambda self: self._view("pass", np.uint32))
    hit_cnt = property(lambda self: self._view("hit", np.uint32))

    def update(self, ox, oy, cx, cy):
        v = lib().orc_grid_update(self._g, _c(ox), _c(oy), len(ox), float(np.asarray(cx).reshape(-1)[0]),
                                  float(np.asarray(cy).reshape(-1)[0]))
        self.visits += v
        return self.pmap

    def occupancy_grid_data(self):
        out = np.empty(self.xw * self.yw, dtype=np.int8)
        lib().orc_occupancy_grid_data(self._g, out)
        return out


def pass_count_threshold(free_inc=0.01, thresh=10.0):
    return lib().orc_pass_count_threshold(free_inc, thresh)


def set_nn_rule(rule):
    """0: candidates ordered by distance (the reference), 1: by the fused square (the device kernels)."""
    lib().orc_set_nn_rule(int(rule))


def nn_rule_splits(reset=True):
    """Queries since the last reset whose nearest neighbour differs between the two orderings."""
    return int(lib().orc_nn_rule_splits(1 if reset else 0))


def replay(ranges, angle_min, angle_max, grid=None, max_iter=30, tolerance=0.001, pose0=(0.0, 0.0, 0.0), threads=1,
           mt_grid=False):
    """ranges float32 [n_scan, n] -> (poses [n_scan-1,3], T [n_scan-1,3,3], iters, visits).
    mt_grid=True uses orc_replay_mt (rays cast in parallel into the integer counters; no datamap)."""
    r = _c(ranges, np.float32)
    n_scan, n = r.shape
    ct, st = trig_tables(angle_min, angle_max, n)
    poses, T, it = np.empty((n_scan - 1, 3)), np.empty((n_scan - 1, 9)), np.empty(n_scan - 1, dtype=np.int32)
    fn = lib().orc_replay_mt if mt_grid else lib().orc_replay
    v = fn(r, n_scan, n, _c(ct), _c(st), max_iter, tolerance, _c(pose0), grid._g if grid else None, poses, T, it, threads)
    if grid is not None:
        grid.visits += v
    return poses, T.reshape(-1, 3, 3), it, v


def laser_estimation(obstacle, pose, angle_min, angle_increment, total_num):
    """obstacle [2,K] -> virtual scan ranges [total_num] (float64)."""
    out = np.empty(total_num)
    lib().orc_laser_estimation(_c(obstacle[0]), _c(obstacle[1]), obstacle.shape[1], _c(pose), angle_min, angle_increment,
                               total_num, out)
    return out


def map_obstacles(data, width, height, resolution, origin_x, origin_y):
    d = _c(data, np.int8)
    ox, oy = np.empty(width * height), np.empty(width * height)
    k = lib().orc_map_obstacles(d, width, height, resolution, origin_x, origin_y, ox, oy)
    return np.vstack((ox[:k], oy[:k]))
