"""The in-scope part of the W12 ``SLAM_EKF`` node: scan matching + map building.

Mirrors the glue of W12m/slam_ekf.py that sits on the hot path:
``laserCallback`` :63-95 (decimation, first scan, odometry, map building, map
publishing), ``calc_odometry`` :109-113, ``laserToNumpy`` :115-123, ``T2u`` :125-128,
``u2T`` :130-137 and the OccupancyGrid layout of ``publishMap`` :252-275.

Two modes.  ``SLAM_EKF()`` is the hot path alone: ``xEst`` is the dead-reckoned ICP pose
(icp.py:185-190), which is what the EKF's prediction step integrates, and one processed scan
costs ONE library call.  ``SLAM_EKF(landmarks=True)`` is the whole W12 node:
landmark extraction (:79, ``extraction.py``) and the landmark EKF (:86, ``ekf_lm.py``) run on
the host between the device-side scan matching and map building, and the map is cast from the
filter's ``xEst`` exactly as :88-90 do.

One processed scan costs ONE library call: ``slam_replay`` over the (previous, current)
scan pair runs polar->Cartesian, the whole ICP solve, the pose composition, the
world-frame transform and the ray casting on the device.
"""
from __future__ import annotations

import math

import numpy as np

from . import _abi
from .ekf_lm import EKF
from .extraction import Extraction
from .icp import ICP, scan_to_pc
from .mapping import Mapping
from .param import get_param

MAX_LASER_RANGE = 30   # slam_ekf.py:18
STATE_SIZE = 3


class SLAM_EKF:  # noqa: N801 (the reference's class name)
    def __init__(self, context=None, online=False, landmarks=False):
        """``online=True`` is the node of w12-mapping-online: +4 end-point evidence
        (W12o/mapping.py:46), every 6th message (W12o/slam_ekf.py:81) and ray origins taken
        from the last /tf message (``tf_callback``, :71-77,104) instead of xEst."""
        self._ctx = context or _abi.default_context()
        self.online = bool(online)
        self.landmarks = bool(landmarks)
        self.ekf = EKF() if self.landmarks else None
        self.extraction = Extraction() if self.landmarks else None
        self.x_online = 0
        self.y_online = 0
        self.z_online = 0
        self.robot_x = get_param('/slam/robot_x', 0)
        self.robot_y = get_param('/slam/robot_y', 0)
        self.robot_theta = get_param('/slam/robot_theta', 0)
        # map geometry (slam_ekf.py:28-33); the reference has no defaults, the launch file
        # sets 20 / 20 / 0.1 (W12m/launch/mapping.launch:14-16)
        self.map_x_width = get_param('/slam/map_width', 20)
        self.map_y_width = get_param('/slam/map_height', 20)
        self.map_reso = get_param('/slam/map_resolution', 0.1)
        self.map_cellx_width = int(round(self.map_x_width / self.map_reso))
        self.map_celly_width = int(round(self.map_y_width / self.map_reso))
        self.mapping = Mapping(self.map_cellx_width, self.map_celly_width, self.map_reso, context=self._ctx,
                               hit_inc=4.0 if self.online else 20.0)
        self.icp = ICP(context=self._ctx)
        self.sensor_sta = [self.robot_x, self.robot_y, self.robot_theta]
        self.isFirstScan = True
        self.laser_count = 0
        self.xOdom = np.zeros((STATE_SIZE, 1))
        self.xEst = np.array([[float(self.robot_x)], [float(self.robot_y)], [float(self.robot_theta)]])
        self.PEst = np.eye(STATE_SIZE)
        self.map_pub = None
        self.last_map = None
        self._prev_ranges = None
        self._cur_ranges = None
        self._angles = None
        self._tar_cloud = None

    # lazily materialised views of the clouds the reference keeps as attributes
    @property
    def tar_pc(self):
        return [] if self._prev_ranges is None else self._pc(self._prev_ranges)

    @property
    def src_pc(self):
        return [] if self._cur_ranges is None else self._pc(self._cur_ranges)

    def _pc(self, ranges):
        class _M:  # minimal LaserScan view
            pass
        m = _M()
        m.ranges, (m.angle_min, m.angle_max) = ranges, self._angles
        return scan_to_pc(m, clip_inf=True, context=self._ctx)

    def tf_callback(self, msg):
        """W12o/slam_ekf.py:71-77: remember the translation of the last transform."""
        for i in msg.transforms:
            self.x_online = i.transform.translation.x
            self.y_online = i.transform.translation.y
            self.z_online = i.transform.translation.z

    def laserCallback(self, msg):
        self.laser_count += 1
        if self.laser_count < (6 if self.online else 5):           # :65-67 / W12o :81
            return
        self.laser_count = 0
        ranges = np.ascontiguousarray(np.asarray(msg.ranges, dtype=np.float32))
        self._angles = (msg.angle_min, msg.angle_max)
        if self.landmarks:
            return self._landmark_callback(msg, ranges)
        if self.isFirstScan:                                       # :74-78
            self.isFirstScan = False
            self._prev_ranges = ranges
            return
        self._cur_ranges = ranges
        n = ranges.shape[0]
        pair = np.ascontiguousarray(np.stack([self._prev_ranges, ranges]))
        ct, st = _abi.trig_tables(msg.angle_min, msg.angle_max, n)
        pose0 = np.ascontiguousarray(self.xEst[:3, 0].reshape(1, 3))
        pose = np.empty((1, 3))
        T = np.empty((1, 9))
        it = np.zeros(1, dtype=np.int32)
        tol = get_param('/icp/tolerance', 0.001)                   # icp.py:40
        _abi.check(_abi.lib().slam_replay(self._ctx.handle, _abi.ptr(pair), _abi.ptr(ct), _abi.ptr(st), 1, 2, n,
                                          _abi.F64, int(self.icp.max_iter), float(tol), _abi.ptr(pose0),
                                          None if self.online else self.mapping._grid, None, _abi.ptr(pose),
                                          _abi.ptr(T), _abi.ptr(it)))
        if self.online:                                            # W12o :102-104
            centre = np.array([[float(self.x_online), float(self.y_online)]])
            _abi.check(_abi.lib().slam_grid_update_scans(self._ctx.handle, self.mapping._grid, _abi.ptr(ranges),
                                                         _abi.ptr(ct), _abi.ptr(st), _abi.ptr(pose), _abi.ptr(centre),
                                                         1, n))
        self._prev_ranges = ranges                                 # calc_odometry :112
        self._tar_cloud = None
        self.last_T = T.reshape(3, 3)
        self.last_u = self.T2u(self.last_T)
        self.xEst[:3, 0] = pose[0]
        self.icp.sensor_sta = [float(v) for v in pose[0]]
        self.publishMap(self.mapping._fetch_pmap())                # :91

    def _landmark_callback(self, msg, ranges):
        """slam_ekf.py:73-95 in full: extraction -> odometry -> EKF -> map from xEst."""
        np_msg = self.laserToNumpy(msg)                            # :73
        if self.isFirstScan:                                       # :74-78
            self.isFirstScan = False
            self._tar_cloud = np_msg
            self._prev_ranges = ranges
            return
        lm = self.extraction.process(np_msg)                       # :79
        if lm is None:                                             # :80-82: nothing else happens, not even
            return                                                 # the odometry target moves on
        self._cur_ranges = ranges
        u = self.calc_odometry(np_msg)                             # :84
        self._prev_ranges = ranges
        z = self.observation(lm)                                   # :85
        self.xEst, self.PEst = self.ekf.estimate(self.xEst, self.PEst, z, u)   # :86
        self.last_u, self.last_landmarks = u, lm
        n = ranges.shape[0]
        ct, st = _abi.trig_tables(msg.angle_min, msg.angle_max, n)
        pose = np.ascontiguousarray(self.xEst[:3, 0].reshape(1, 3))
        centre = None
        if self.online:
            centre = np.array([[float(self.x_online), float(self.y_online)]])
        _abi.check(_abi.lib().slam_grid_update_scans(self._ctx.handle, self.mapping._grid, _abi.ptr(ranges), _abi.ptr(ct),
                                                     _abi.ptr(st), _abi.ptr(pose), _abi.ptr(centre), 1, n))   # :88-90
        self.publishMap(self.mapping._fetch_pmap())                # :91

    def observation(self, lm):
        """slam_ekf.py:96-106: landmarks (sensor frame) -> rows (range, bearing, index)."""
        z = np.zeros((0, 3))
        for i in range(len(lm.id)):
            dx, dy = lm.position_x[i], lm.position_y[i]
            zi = np.array([math.hypot(dx, dy), self.ekf.pi_2_pi(math.atan2(dy, dx)), i])
            z = np.vstack((z, zi))
        return z

    def calc_odometry(self, np_msg):
        """slam_ekf.py:109-113 on explicit clouds (3xN): returns u = [tx, ty, dyaw]^T and
        makes ``np_msg`` the next target."""
        tar = self._tar_cloud if self._tar_cloud is not None else self.tar_pc
        T = self.icp.process(tar, np_msg)
        self._tar_cloud = np_msg
        return self.T2u(T)

    def laserToNumpy(self, msg):
        return scan_to_pc(msg, clip_inf=True, context=self._ctx)   # :115-123

    def T2u(self, t):
        dw = math.atan2(t[1, 0], t[0, 0])                          # :126
        return np.array([[t[0, 2], t[1, 2], dw]]).T

    def u2T(self, u):
        dx, dy, w = float(u[0]), float(u[1]), float(u[2])          # :131-133
        return np.array([[math.cos(w), -math.sin(w), dx], [math.sin(w), math.cos(w), dy]])

    def publishMap(self, pmap):
        """OccupancyGrid content of slam_ekf.py:252-275; ``data`` comes from the device in
        the wire layout (data[y*width + x] = int8(pmap[x][y]), :270-271)."""
        self.last_map = {
            "frame_id": "map", "resolution": self.map_reso,
            "width": self.map_cellx_width, "height": self.map_celly_width,
            "origin": (-self.map_cellx_width * self.map_reso / 2.0, -self.map_celly_width * self.map_reso / 2.0, 0.0),
            "data": self.mapping.occupancy_grid_data(),
        }
        if self.map_pub is not None:
            self.map_pub.publish(self.last_map)
