"""``Localization``: scan-to-map observation on the GPU, with the reference's class surface.

Mirrors ``Localization`` of W9 = "W9_Fusion Localization (LiDAR Odometry)/course_agv_slam/
scripts/localization.py":

* ``updateMap(msg)``             :54-60    obstacle cells of an OccupancyGrid  -> ``slam_map_obstacles``
* ``laserEstimation(msg, x)``    :128-150  the scan the map would give at x    -> ``slam_virtual_scan``
* ``laserToNumpy(msg)``          :168-174  ranges -> 3xN points                -> ``slam_scan_to_points[_f64]``
* ``calc_map_observation(msg)``  :152-157  virtual scan + ICP against the scan -> ``slam_map_observation``
* ``calc_odometry(msg)``         :159-166  ICP against the previous scan       -> ``slam_icp_batch``
* ``laserCallback(msg)``         :66-126   the three ICP calls of one processed scan + pose algebra

``map_observation_batch`` is the many-hypotheses form (one virtual scan + ICP per pose in one
launch each), which is what the projection's data parallelism is for.

The 3x3 filter of the reference (``ekf.py``) is a pluggable hook (``self.ekf``, any object with
``estimate(xEst, PEst, z, T)``); :class:`EKF` below restates it in a dozen numpy lines, since
it is host control logic and not part of the accelerated path.  No ROS is needed: publishers
are optional hooks as in :class:`ICP`.
"""
from __future__ import annotations

import copy
import math

import numpy as np

from . import _abi
from .icp import ICP, scan_to_pc
from .param import get_param


class EKF:
    """Pose filter of W9/ekf.py:17-87: odometry T predicts, the absolute pose z corrects.
    (The reference adds its ``R`` in the prediction and its ``Q`` in the innovation; both
    are the same diagonal, ekf.py:6-13.)"""

    NOISE = np.diag([0.2, 0.2, math.pi / 60]) ** 2

    def odom_model(self, x, T):
        yaw = math.atan2(T[1, 0], T[0, 0])
        c, s = math.cos(x[2]), math.sin(x[2])
        return np.array([x[0] + c * T[0, 2] - s * T[1, 2], x[1] + s * T[0, 2] + c * T[1, 2], x[2] + yaw], dtype=float)

    def jacob_f(self, x, T):
        J = np.eye(3)
        J[0, 2] = -T[0, 2] * T[1, 0] - T[1, 2] * T[0, 0]
        J[1, 2] = T[0, 2] * T[0, 0] - T[1, 2] * T[1, 0]
        return J

    def estimate(self, xEst, PEst, z, T):
        x_pred = self.odom_model(xEst, T)
        J = self.jacob_f(xEst, T)
        P_pred = J.dot(PEst).dot(J.T) + self.NOISE
        K = P_pred.dot(np.linalg.inv(P_pred + self.NOISE))
        x_new = x_pred + K.dot(np.asarray(z, dtype=float) - x_pred)
        return x_new, (np.eye(3) - K).dot(P_pred)


def _compose(s, T):
    """The pose algebra repeated at localization.py:79-83, 102-106, 113-118."""
    yaw = math.atan2(T[1, 0], T[0, 0])
    return [s[0] + math.cos(s[2]) * T[0, 2] - math.sin(s[2]) * T[1, 2],
            s[1] + math.sin(s[2]) * T[0, 2] + math.cos(s[2]) * T[1, 2],
            s[2] + yaw]


class Localization:
    def __init__(self, context=None, ekf=None):
        self._ctx = context or _abi.default_context()
        self.icp = ICP(context=self._ctx)
        self.ekf = ekf if ekf is not None else EKF()
        self.laser_count = 0
        self.robot_x = get_param('/icp/robot_x', 0)            # localization.py:22-25
        self.robot_y = get_param('/icp/robot_y', 0)
        self.robot_theta = get_param('/icp/robot_theta', 0)
        self.sensor_sta = [self.robot_x, self.robot_y, self.robot_theta]
        self.isFirstScan = True
        self.src_pc = []
        self.tar_pc = []
        self.xOdom = [0, 0, 0]
        self.xEst = [0, 0, 0]
        self.PEst = np.eye(3)
        self.obstacle = np.zeros((2, 0))
        self.obstacle_r = 10
        self.new = [0, 0, 0]
        self.new2 = [0, 0, 0]
        self.map = None
        self.target_laser = None
        self.laser_pub = self.location_pub = self.location_pub1 = self.odom_pub = None
        self.odom_broadcaster = None
        self.last_T = None
        self.last_t = None

    # ------------------------------------------------------------------ map side
    def updateMap(self, msg):
        """OccupancyGrid message (``data`` in wire order [y*width + x], ``info.width /
        height / resolution / origin.position``) -> ``self.obstacle`` 2xK (:54-60).

        The reference reshapes with ``(-1, info.height)`` and transposes, which reads the
        wire order correctly only for square maps (all of the reference's are); the same
        interpretation is kept: cell c of ``data`` is (tx, ty) = (c % height, c // height).
        The obstacle list comes back sorted like ``np.nonzero`` (tx-major)."""
        self.map = msg
        info = msg.info
        data = np.ascontiguousarray(np.asarray(msg.data, dtype=np.int8))
        h = int(info.height)
        if h <= 0 or data.size % h:
            raise ValueError("cannot reshape array of size %d into shape (-1,%d)" % (data.size, h))
        rows = data.size // h
        res = float(info.resolution)
        ox0, oy0 = float(info.origin.position.x), float(info.origin.position.y)
        cap = int(data.size)
        ox = np.empty(cap)
        oy = np.empty(cap)
        import ctypes as C
        k = C.c_int(0)
        # data viewed as [rows][h] with tx = column: "wire" with width := h, height := rows
        _abi.check(_abi.lib().slam_map_obstacles(self._ctx.handle, _abi.ptr(data), h, rows, 1, res, ox0, oy0,
                                                 _abi.ptr(ox), _abi.ptr(oy), cap, C.byref(k)))
        k = k.value
        ox, oy = ox[:k], oy[:k]
        order = np.lexsort((oy, ox))                      # np.nonzero order: tx, then ty
        self.obstacle = np.vstack((ox[order], oy[order]))
        self.obstacle_r = info.resolution

    def _poses(self, x, B=None):
        """laserEstimation takes the POSITION from self.xEst and only the heading from its
        argument x (:138-139)."""
        return np.array([[float(self.xEst[0]), float(self.xEst[1]), float(x[2])]], dtype=np.float64)

    def virtual_ranges(self, msg, poses):
        """poses [B,3] -> float64 ranges [B,n] (``slam_virtual_scan``)."""
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(-1, 3))
        n = len(msg.ranges)
        obs = np.ascontiguousarray(np.asarray(self.obstacle, dtype=np.float64).reshape(2, -1))
        out = np.empty((poses.shape[0], n))
        _abi.check(_abi.lib().slam_virtual_scan(self._ctx.handle, _abi.ptr(obs[0]), _abi.ptr(obs[1]), obs.shape[1],
                                                _abi.ptr(poses), poses.shape[0], float(msg.angle_min),
                                                float(msg.angle_increment), n, _abi.ptr(out)))
        return out

    def laserEstimation(self, msg, x):
        """A copy of msg whose ranges are what the map predicts at (xEst.x, xEst.y, x[2])
        (:128-150)."""
        data = copy.copy(msg)
        data.ranges = self.virtual_ranges(msg, self._poses(x))[0].tolist()
        self.target_laser = data
        return data

    def laserToNumpy(self, msg):
        """ranges -> 3xN [x; y; 1] (:168-174).  float32-representable ranges (a wire
        message) take the float32 entry point; a float64 virtual scan the float64 one."""
        r64 = np.ascontiguousarray(np.asarray(msg.ranges, dtype=np.float64))
        if np.array_equal(r64.astype(np.float32).astype(np.float64), r64, equal_nan=True):
            return scan_to_pc(msg, clip_inf=False, context=self._ctx)
        n = r64.shape[0]
        ct, st = _abi.trig_tables(msg.angle_min, msg.angle_max, n)
        pts = np.empty((2, n))
        _abi.check(_abi.lib().slam_scan_to_points_f64(self._ctx.handle, _abi.ptr(r64), _abi.ptr(ct), _abi.ptr(st), 1, n,
                                                      _abi.ptr(pts)))
        pc = np.ones([3, n])
        pc[0:2, :] = pts
        return pc

    # ------------------------------------------------------------------ observation
    def map_observation_batch(self, msg, poses, src_pc=None):
        """Virtual scan + ICP for every pose hypothesis in one call (``slam_map_observation``):
        poses [B,3] -> (T [B,3,3], iters [B]).  ``src_pc`` 3xN (default: self.src_pc)."""
        poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(-1, 3))
        B = poses.shape[0]
        n = len(msg.ranges)
        src = np.ascontiguousarray(np.asarray(self.src_pc if src_pc is None else src_pc, dtype=np.float64)[:2, :])
        if src.shape[1] != n:
            raise ValueError("scan has %d beams but src_pc has %d points" % (n, src.shape[1]))
        obs = np.ascontiguousarray(np.asarray(self.obstacle, dtype=np.float64).reshape(2, -1))
        ct, st = _abi.trig_tables(msg.angle_min, msg.angle_max, n)
        T = np.empty((B, 9))
        it = np.zeros(B, dtype=np.int32)
        tol = get_param('/icp/tolerance', 0.001)
        _abi.check(_abi.lib().slam_map_observation(self._ctx.handle, _abi.ptr(obs[0]), _abi.ptr(obs[1]), obs.shape[1],
                                                   _abi.ptr(poses), _abi.ptr(src), B, n, 1, _abi.ptr(ct), _abi.ptr(st),
                                                   float(msg.angle_min), float(msg.angle_increment),
                                                   int(self.icp.max_iter), float(tol), _abi.ptr(T), _abi.ptr(it)))
        if not np.all(np.isfinite(T)):
            raise np.linalg.LinAlgError("SVD did not converge")
        return T.reshape(B, 3, 3), it

    def calc_map_observation(self, msg):
        """:152-157."""
        T, _ = self.map_observation_batch(msg, self._poses(self.xEst))
        return T[0]

    def calc_odometry(self, msg):
        """:159-166 (the very first target is the MAP's virtual scan, not a real one)."""
        if self.isFirstScan:
            self.tar_pc = self.laserToNumpy(self.laserEstimation(msg, self.xEst))
            self.isFirstScan = False
            self.laser_count = 0
        self.src_pc = self.laserToNumpy(msg)
        T = self.icp.process(self.tar_pc, self.src_pc)
        self.tar_pc = self.src_pc
        return T

    def laserCallback(self, msg):
        """:66-126: every 6th message -> odometry (twice, as the reference does: the second
        call matches the scan against itself) + map observation + filter."""
        self.laser_count += 1
        if self.laser_count <= 5:
            return
        self.laser_count = 0
        T = self.calc_odometry(msg)
        self.xOdom = _compose(self.xOdom, T)
        T = self.calc_odometry(msg)                       # :100
        self.new = _compose(self.xEst, T)
        t = self.calc_map_observation(msg)
        self.new2 = _compose(self.xEst, t)
        self.last_T, self.last_t = T, t
        self.xEst, self.PEst = self.ekf.estimate(self.xEst, self.PEst, self.new2, T)
        self.publishResult()

    def publishResult(self):
        """Pose content of :175-245 handed to whichever hooks are set."""
        def odom(s, child):
            q = (0.0, 0.0, math.sin(s[2] / 2.0), math.cos(s[2] / 2.0))
            o = {"frame_id": "world_base", "child_frame_id": child, "position": (s[0], s[1], 0.001), "orientation": q}
            if self.odom_broadcaster is not None:
                self.odom_broadcaster.sendTransform((s[0], s[1], 0.001), q, None, child, "world_base")
            return o
        self.last_odom = {"ekf_w8": odom(self.xEst, "ekf_w8"), "ekf_w9": odom(self.xEst, "ekf_w9"),
                          "icp_odom": odom(self.xOdom, "icp_odom")}
        for pub, key in ((self.location_pub, "ekf_w8"), (self.location_pub1, "ekf_w9"), (self.odom_pub, "icp_odom")):
            if pub is not None:
                pub.publish(self.last_odom[key])
        if self.laser_pub is not None and self.target_laser is not None:
            self.laser_pub.publish(self.target_laser)
