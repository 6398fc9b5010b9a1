"""``ICP`` scan matcher on the GPU, with the reference's class surface.

Mirrors the two generations of the reference class:
* the library form ``ICP.process(tar_pc, src_pc) -> T`` of
  W12m/icp.py:38-88 (``findNearest`` :90-114, ``getTransform`` :149-179,
  ``laserToNumpy`` :216-229, ``publishResult`` :181-212), and
* the node form ``ICP.laserCallback(msg)`` of W7/icp.py:39-98, whose arithmetic is the
  W12 generation's (W7's own cannot run: NameError at :118, IndexError at :136).

Every method that computes is one call into libslamhip (``slam_icp_batch``, ``slam_nn``,
``slam_kabsch2d``, ``slam_scan_to_points``, ``slam_pose_compose``): the whole ICP loop of
``process`` runs inside one kernel launch.  There is no host implementation.

ROS: none is needed.  ``odom_pub`` / ``odom_broadcaster`` are optional hooks; when left
``None`` the pose that would have been published is kept in ``last_odom``.
"""
from __future__ import annotations

import math

import numpy as np

from . import _abi
from .param import get_param


def _check_finite(T):
    """The reference takes numpy.linalg.svd of W (icp.py:161), which raises LinAlgError when a
    NaN / inf coordinate (e.g. an unclipped inf range, W7 icp.py:192-194) reached it; the
    device propagates the NaN into T instead, so the same exception is raised here."""
    if not np.all(np.isfinite(T)):
        raise np.linalg.LinAlgError("SVD did not converge")


def _soa(points_nx2):
    """[N,2] (any strides, e.g. a transposed view) -> contiguous float64 [2,N]."""
    a = np.asarray(points_nx2, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("expected an N x 2 array, got shape %r" % (a.shape,))
    return np.ascontiguousarray(a.T)


class ICP:
    def __init__(self, context=None):
        self.laser_count = 0
        # robot init states (icp.py:14-18)
        self.robot_x = get_param('/icp/robot_x', 0)
        self.robot_y = get_param('/icp/robot_y', 0)
        self.robot_theta = get_param('/icp/robot_theta', 0)
        self.sensor_sta = [self.robot_x, self.robot_y, self.robot_theta]
        self.max_iter = get_param('/icp/max_iter', 30)         # icp.py:21
        self.dis_th = get_param('/icp/dis_th', 5)              # read, never used (icp.py:23)
        self.tolerance = get_param('/icp/tolerance', 0.001)    # icp.py:25
        self.isFirstScan = True
        self.src_pc = []
        self.tar_pc = []
        self.odom_pub = None
        self.odom_broadcaster = None
        self.last_odom = None
        self.last_iters = 0
        self.last_mean_error = 0.0
        self._ctx = context or _abi.default_context()

    # ------------------------------------------------------------------ library form
    def process(self, tar_pc, src_pc):
        """3xN target, 3xN source (rows x, y, 1; N may differ) -> 3x3 T mapping the
        source frame into the target frame (icp.py:38-88).  As in the reference the
        tolerance is re-read from the parameter table on every call (:40)."""
        tolerance = get_param('/icp/tolerance', 0.001)
        return self._solve(tar_pc, src_pc, self.max_iter, tolerance)

    def _solve(self, tar_pc, src_pc, max_iter, tolerance):
        tar = np.ascontiguousarray(np.asarray(tar_pc, dtype=np.float64)[:2, :])
        src = np.ascontiguousarray(np.asarray(src_pc, dtype=np.float64)[:2, :])
        T = np.empty((3, 3), dtype=np.float64)
        it = np.zeros(1, dtype=np.int32)
        err = np.zeros(1, dtype=np.float64)
        _abi.check(_abi.lib().slam_icp_batch(self._ctx.handle, _abi.ptr(tar), _abi.ptr(src), 1, tar.shape[1],
                                             src.shape[1], _abi.F64, 0, 0, None, int(max_iter), float(tolerance),
                                             _abi.ptr(T), _abi.ptr(it), _abi.ptr(err)))
        self.last_iters, self.last_mean_error = int(it[0]), float(err[0])
        _check_finite(T)
        return T

    def findNearest(self, src, tar):
        """src N x 2, tar M x 2 -> (distances[N], indices[N]); lowest index wins ties
        (icp.py:90-114)."""
        s, t = _soa(src), _soa(tar)
        n = s.shape[1]
        distances = np.zeros(n)
        idx = np.zeros(n, dtype=np.int32)
        _abi.check(_abi.lib().slam_nn(self._ctx.handle, _abi.ptr(s), _abi.ptr(t), 1, n, t.shape[1], _abi.F64,
                                      _abi.ptr(distances), _abi.ptr(idx)))
        return distances, idx.astype(np.int64)

    def getTransform(self, src, tar):
        """Paired N x 2 rows -> 3x3 rigid T (icp.py:149-179)."""
        s, t = _soa(src), _soa(tar)
        if s.shape != t.shape:
            raise ValueError("operands could not be broadcast together with shapes %r %r" % (s.T.shape, t.T.shape))
        T = np.empty((3, 3), dtype=np.float64)
        _abi.check(_abi.lib().slam_kabsch2d(self._ctx.handle, _abi.ptr(s), _abi.ptr(t), 1, s.shape[1], _abi.ptr(T)))
        _check_finite(T)
        return T

    def laserToNumpy(self, msg):
        """LaserScan -> 3xN [x; y; 1] (icp.py:216-229; no inf/NaN handling here)."""
        return scan_to_pc(msg, clip_inf=False, context=self._ctx)

    # ------------------------------------------------------------------ node form
    def laserCallback(self, msg):
        """W7/icp.py:39-98: first scan becomes the target; then every 6th message is
        matched against the previous processed one and the pose is dead-reckoned."""
        if self.isFirstScan:
            self.tar_pc = self.laserToNumpy(msg)
            self.isFirstScan = False
            self.laser_count = 0
            return
        self.laser_count += 1
        if self.laser_count <= 5:
            return
        self.laser_count = 0
        self.src_pc = self.laserToNumpy(msg)
        T = self._solve(self.tar_pc, self.src_pc, self.max_iter, self.tolerance)   # :74-92
        self.tar_pc = self.src_pc                                                  # :95
        self.publishResult(T)                                                      # :96

    def publishResult(self, T):
        """Pose part of icp.py:181-212: compose T onto ``sensor_sta`` (theta is not
        wrapped), then hand the Odometry / TF content to the hooks."""
        T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(1, 9))
        pose0 = np.array([[float(v) for v in self.sensor_sta]], dtype=np.float64)
        out = np.empty((1, 3), dtype=np.float64)
        _abi.check(_abi.lib().slam_pose_compose(self._ctx.handle, _abi.ptr(T), _abi.ptr(pose0), 1, 1, _abi.ptr(out)))
        self.sensor_sta[0], self.sensor_sta[1], self.sensor_sta[2] = float(out[0, 0]), float(out[0, 1]), float(out[0, 2])
        s = self.sensor_sta
        q = (0.0, 0.0, math.sin(s[2] / 2.0), math.cos(s[2] / 2.0))   # quaternion_from_euler(0, 0, yaw)
        self.last_odom = {"frame_id": "world_base", "child_frame_id": "icp_odom",
                          "position": (s[0], s[1], 0.001), "orientation": q}
        if self.odom_broadcaster is not None:
            self.odom_broadcaster.sendTransform((s[0], s[1], 0.001), q, None, "icp_odom", "world_base")
        if self.odom_pub is not None:
            self.odom_pub.publish(self.last_odom)

    def calcDist(self, a, b):
        return math.hypot(a[0] - b[0], a[1] - b[1])


def scan_to_pc(msg, clip_inf, context=None, dtype="f64"):
    """LaserScan duck-type -> 3xN float64 [x; y; 1] via ``slam_scan_to_points``.
    ``clip_inf`` selects the SLAM_EKF variant (inf -> 30 m, W12m/slam_ekf.py:119)."""
    ctx = context or _abi.default_context()
    ranges = np.ascontiguousarray(np.asarray(msg.ranges, dtype=np.float32))
    n = ranges.shape[0]
    ct, st = _abi.trig_tables(msg.angle_min, msg.angle_max, n)
    code = _abi.DTYPES[dtype]
    pts = np.empty((2, n), dtype=_abi.NP_DTYPES[code])
    _abi.check(_abi.lib().slam_scan_to_points(ctx.handle, _abi.ptr(ranges), _abi.ptr(ct), _abi.ptr(st), 1, n,
                                              int(bool(clip_inf)), code, _abi.ptr(pts)))
    pc = np.ones([3, n])
    pc[0:2, :] = pts
    return pc
