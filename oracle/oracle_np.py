"""CPU oracle (NumPy / pure Python) for the ICP + occupancy-grid hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and there only as the checker.

This file restates, in plain float64 Python/NumPy, the algorithm of the reference
(paths relative to /root/reference):

  W12m = "W12_LiDAR SLAM/w12-mapping/course_agv_slam/scripts"
  W7   = "W7_Dead Reckoning (ICP)/course_agv_slam/scripts"

Each function cites the reference lines it follows.  Parity is PINNED: the functions
here are checked against golden vectors produced by importing the reference's own
code (oracle/gen_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).

The ``*_loop`` functions keep the reference's loop structure (slow, small inputs
only); the unsuffixed ones are vectorised equivalents used on larger inputs and are
themselves checked against the loop forms.
"""
from __future__ import annotations

import math

import numpy as np

MAX_LASER_RANGE = 30  # W12m/slam_ekf.py:18


# ----------------------------------------------------------------------------
# a-7  polar -> Cartesian
# ----------------------------------------------------------------------------
def laser_to_numpy(ranges, angle_min, angle_max, clip_inf=False):
    """W7/icp.py:182-195 (clip_inf=False) and W12m/slam_ekf.py:115-123
    (clip_inf=True: inf -> 30 m at :119; the NaN line :120 is a no-op because
    ``x == nan`` is never true)."""
    total_num = len(ranges)
    pc = np.ones([3, total_num])
    range_l = np.array(ranges, dtype=np.float64)
    if clip_inf:
        range_l[range_l == np.inf] = MAX_LASER_RANGE
    angle_l = np.linspace(angle_min, angle_max, total_num)
    pc[0, :] = np.cos(angle_l) * range_l
    pc[1, :] = np.sin(angle_l) * range_l
    return pc


# ----------------------------------------------------------------------------
# a-4  brute-force nearest neighbour
# ----------------------------------------------------------------------------
def find_nearest_loop(src, tar):
    """W12m/icp.py:90-114.  src [N,2], tar [M,2].  Strict ``<`` keeps the lowest j on
    ties (:103); a NaN distance never wins, leaving (distance 0, index 0) (:96-97)."""
    n = src.shape[0]
    indices = np.zeros(n, dtype=np.int64)
    distances = np.zeros(n)
    for i in range(n):
        min_dist = np.inf
        sx, sy = src[i, 0], src[i, 1]
        for j in range(tar.shape[0]):
            # the reference's own expression (:102): sqrt(x.dot(x)), whose two-element dot is BLAS
            # arithmetic - fma(x1, x1, x0*x0) in this container's NumPy, tests/test_nn_near_ties.py
            dist = float(np.linalg.norm(np.array([sx - tar[j, 0], sy - tar[j, 1]])))
            if dist < min_dist:
                min_dist = dist
                indices[i] = j
                distances[i] = dist
    return distances, indices


def find_nearest(src, tar):
    """Vectorised form of :func:`find_nearest_loop` (argmin returns the first minimum,
    i.e. the lowest j, matching the strict ``<`` of icp.py:103)."""
    dx = src[:, 0:1] - tar[None, :, 0]
    dy = src[:, 1:2] - tar[None, :, 1]
    d = np.sqrt(dx * dx + dy * dy)            # unfused: the last place may differ from the reference's norm
    dd = np.where(np.isnan(d), np.inf, d)
    idx = np.argmin(dd, axis=1)
    dist = dd[np.arange(src.shape[0]), idx]
    never = ~np.isfinite(dist)  # inf < inf is False: nothing ever won
    idx = np.where(never, 0, idx)
    dist = np.where(never, 0.0, dist)
    # candidates within a few units in the last place of the minimum are decided - and every
    # distance is produced - by the reference's own expression, as find_nearest_loop does
    close = dd <= (dist * (1.0 + 1e-14))[:, None]
    for i in range(src.shape[0]):
        if never[i]:
            continue
        best, bj = np.inf, 0
        for j in np.nonzero(close[i])[0]:
            dn = float(np.linalg.norm(np.array([src[i, 0] - tar[j, 0], src[i, 1] - tar[j, 1]])))
            if dn < best:
                best, bj = dn, j
        idx[i], dist[i] = bj, best
    return dist, idx.astype(np.int64)


# ----------------------------------------------------------------------------
# a-5  2-D Kabsch via 2x2 SVD
# ----------------------------------------------------------------------------
def collapsed(rows):
    """True when every row of the set is ONE point (bitwise).  Then W = BB^T.AA is
    mathematically zero and every rotation is optimal; the reference's centred rows are
    rounding noise instead (np.mean of n equal values is not that value), W ~ 1e-31, and the
    rotation its SVD returns is arbitrary.  DOCUMENTED DEVIATION: the oracles and the product
    return the canonical R = I, t = centroid_B - centroid_A (the SVD of an exact zero matrix);
    tests/golden/g8_collapsed.npz records what the reference itself returns."""
    rows = np.asarray(rows)
    return rows.shape[0] > 0 and bool(np.all(rows == rows[0]))


def get_transform(src, tar, collapsed_rule="canonical"):
    """W12m/icp.py:149-179.  src, tar [N,2] paired rows -> T 3x3.  Uses the
    ``Vt[1,:]`` reflection fix of the W12 generation (:168); W7/icp.py:136 indexes
    ``Vt[2,:]`` and cannot run.  ``collapsed_rule="reference"`` is the reference's own
    arithmetic on collapsed sets too (the SVD of its rounding noise, see collapsed()):
    the mode that keeps this restatement independent of the documented deviation and is
    checked against tests/golden/g8_collapsed.npz."""
    centroid_a = np.mean(src, axis=0)
    centroid_b = np.mean(tar, axis=0)
    aa = src - centroid_a
    bb = tar - centroid_b
    w = np.dot(bb.transpose(), aa)
    if collapsed_rule != "reference" and (collapsed(src) or collapsed(tar)):
        w = np.zeros((2, 2))          # documented canonical answer, see collapsed()
    u, _s, vt = np.linalg.svd(w)
    r = np.dot(u, vt)
    if np.linalg.det(r) < 0:
        vt[1, :] *= -1
        r = np.dot(u, vt)
    t = centroid_b - np.dot(r, centroid_a)
    out = np.identity(3)
    out[:2, :2] = r
    out[0, 2] = t[0]
    out[1, 2] = t[1]
    return out


def get_transform_closed_form(src, tar):
    """Closed form of :func:`get_transform`: the rotation that
    maximises tr(R W^T) is R = rot(atan2(W10 - W01, W00 + W11)).  This is what the
    HIP kernel evaluates; tests check it against the SVD form."""
    ca = np.mean(src, axis=0)
    cb = np.mean(tar, axis=0)
    aa = src - ca
    bb = tar - cb
    w = np.dot(bb.transpose(), aa)
    if collapsed(src) or collapsed(tar):
        w = np.zeros((2, 2))
    a = w[0, 0] + w[1, 1]
    b = w[1, 0] - w[0, 1]
    h = math.hypot(a, b)
    c, s = (1.0, 0.0) if h == 0.0 else (a / h, b / h)
    out = np.identity(3)
    out[0, 0], out[0, 1], out[1, 0], out[1, 1] = c, -s, s, c
    out[0, 2] = cb[0] - (c * ca[0] - s * ca[1])
    out[1, 2] = cb[1] - (s * ca[0] + c * ca[1])
    return out


# ----------------------------------------------------------------------------
# a-3  ICP.process
# ----------------------------------------------------------------------------
def icp_process(tar_pc, src_pc, max_iter=30, tolerance=0.001, nn=find_nearest, return_info=False):
    """W12m/icp.py:38-88.  tar_pc, src_pc [3,N] (rows x, y, 1) -> T [3,3] mapping the
    source frame into the target frame.  Loop: NN (:67) -> Kabsch on matched pairs
    (:69) -> src = T.src (:71) -> mean of the PRE-update distances (:75) -> break when
    |pre - mean| < tol (:76-77; pre starts at 0) -> final T from the original source to
    the moved source (:81)."""
    b = np.array(tar_pc[:2, :], dtype=np.float64)
    tar = np.ones((3, b.shape[1]))
    tar[:2, :] = b
    a = np.array(src_pc[:2, :], dtype=np.float64)
    src = np.ones((3, a.shape[1]))
    src[:2, :] = a
    pre_error = 0.0
    iter_cnt = 0
    mean_error = 0.0
    for _ in range(max_iter):
        distances, indices = nn(src[:2, :].transpose(), tar[:2, :].transpose())
        t = get_transform(src[:2, :].transpose(), tar[:2, indices].transpose())
        src = np.dot(t, src)
        iter_cnt += 1
        mean_error = np.sum(distances) / distances.size
        if abs(pre_error - mean_error) < tolerance:
            break
        pre_error = mean_error
    out = get_transform(a.transpose(), src[:2, :].transpose())
    if return_info:
        return out, iter_cnt, float(mean_error)
    return out


# ----------------------------------------------------------------------------
# a-6 / a-8  pose glue
# ----------------------------------------------------------------------------
def compose_pose(sta, t):
    """W7/icp.py:153-158 (= W12m/icp.py:185-190).  sta = [x, y, theta]; theta is not
    wrapped."""
    delta_yaw = math.atan2(t[1, 0], t[0, 0])
    x = sta[0] + math.cos(sta[2]) * t[0, 2] - math.sin(sta[2]) * t[1, 2]
    y = sta[1] + math.sin(sta[2]) * t[0, 2] + math.cos(sta[2]) * t[1, 2]
    return [x, y, sta[2] + delta_yaw]


def t2u(t):
    """W12m/slam_ekf.py:125-128."""
    return np.array([[t[0, 2], t[1, 2], math.atan2(t[1, 0], t[0, 0])]]).T


def u2t(u):
    """W12m/slam_ekf.py:130-137 (2x3)."""
    dx, dy, w = float(u[0]), float(u[1]), float(u[2])
    return np.array([[math.cos(w), -math.sin(w), dx], [math.sin(w), math.cos(w), dy]])


def world_points(pose, pc):
    """W12m/slam_ekf.py:89: obs = u2T(pose) . pc  (2x3 . 3xN)."""
    return u2t(pose).dot(pc)


# ----------------------------------------------------------------------------
# a-11  float-error Bresenham
# ----------------------------------------------------------------------------
def bresenham_path(start, end):
    """W12m/bresenham.py:2-58.  Integer endpoints -> list of (x, y) from start to end
    inclusive; identical endpoints give an empty list (:10-11).  The error term is a
    float64 accumulated with ``error += dy/float(dx)`` (:35,:51), NOT integer
    Bresenham; 15 % of lines differ from the integer algorithm."""
    x0, y0 = int(start[0]), int(start[1])
    x1, y1 = int(end[0]), int(end[1])
    path = []
    if x0 == x1 and y0 == y1:
        return path
    steep = abs(y1 - y0) > abs(x1 - x0)
    if steep:
        x0, y0 = y0, x0
        x1, y1 = y1, x1
    flag = 0
    if x0 > x1:
        flag = 1
        x0, x1 = x1, x0
        y0, y1 = y1, y0
    dx = x1 - x0
    dy = abs(y1 - y0)
    error = 0.0
    derr = dy / float(dx)
    y = y0
    ystep = 1 if y0 < y1 else -1
    for x in range(x0, x1 + 1):
        path.append((y, x) if steep else (x, y))
        error += derr
        if error >= 0.5:
            y += ystep
            error -= 1.0
    if flag == 1:
        path.reverse()
    return path


# ----------------------------------------------------------------------------
# a-9 / a-10  Mapping
# ----------------------------------------------------------------------------
def pass_count_threshold(free_inc=0.01, thresh=10.0, base=0.0):
    """Smallest k for which the float64 running sum ``base`` + k additions of ``free_inc``
    exceeds ``thresh`` (mapping.py:43,47): 1001 for (0.01, 10) because the
    sequential sum of 1000 x 0.01 is 9.99999999999983."""
    acc, k = base, 0
    while not acc > thresh:
        acc += free_inc
        k += 1
    return k


def occupied_rule(free_inc=0.01, hit_inc=20.0, thresh=10.0):
    """The canonical integer rule behind ``pmap`` (include/slam_hip.h, slam_grid_create): a
    cell with h hits and p passes is occupied iff h >= len(table) or p >= table[h], where the
    float sum is taken hits first.  [1001] for the reference's +20; [1001, 601, 201]-ish for
    the +4 of w12-mapping-online (W12o/mapping.py:46), whose own answer depends on the order
    of arrival when p is exactly table[h] - 1 or table[h]."""
    table, base = [], 0.0
    while not base > thresh:
        table.append(pass_count_threshold(free_inc, thresh, base))
        base += hit_inc
    return table


class Mapping:
    """W12m/mapping.py:8-51 with the index rule generalised to
    ``int(scale * (x + offset))``; the reference hard-codes scale = offset = 10
    (:33-36) whatever ``xyreso`` is, which is the default here."""

    def __init__(self, xw, yw, xyreso, scale=10.0, offset_x=10.0, offset_y=10.0,
                 free_inc=0.01, hit_inc=20.0, thresh=10.0):
        self.xw, self.yw, self.xyreso = xw, yw, xyreso
        self.scale, self.offset_x, self.offset_y = scale, offset_x, offset_y
        self.free_inc, self.hit_inc, self.thresh = free_inc, hit_inc, thresh
        self.pmap = 50 * np.ones((xw, yw))        # :14
        self.datamap = np.zeros((xw, yw))         # :15
        self.pass_cnt = np.zeros((xw, yw), dtype=np.int64)   # integer restatement of the
        self.hit_cnt = np.zeros((xw, yw), dtype=np.int64)    # same evidence (:42-45)

    def update(self, ox, oy, center_x, center_y):
        """mapping.py:22-51: per beam (skipped when ox is inf, :30) truncate the four
        world coordinates toward zero (:33-36), rasterise centre -> endpoint (:38),
        and for every in-bounds cell add ``free_inc``, or ``hit_inc`` on the last
        cell of the path (:41-45), then re-threshold that cell (:47-50)."""
        center_x = float(np.asarray(center_x).reshape(-1)[0])
        center_y = float(np.asarray(center_y).reshape(-1)[0])
        for i in range(len(ox)):
            if np.isinf(ox[i]):
                continue
            px_o = int(self.scale * (ox[i] + self.offset_x))
            py_o = int(self.scale * (oy[i] + self.offset_y))
            px_c = int(self.scale * (center_x + self.offset_x))
            py_c = int(self.scale * (center_y + self.offset_y))
            path = bresenham_path([px_c, py_c], [px_o, py_o])
            last = len(path) - 1
            for j, (lpx, lpy) in enumerate(path):
                if 0 <= lpx < self.xw and 0 <= lpy < self.yw:
                    if j < last:
                        self.datamap[lpx][lpy] += self.free_inc
                        self.pass_cnt[lpx][lpy] += 1
                    else:
                        self.datamap[lpx][lpy] += self.hit_inc
                        self.hit_cnt[lpx][lpy] += 1
                    self.pmap[lpx][lpy] = 100 if self.datamap[lpx][lpy] > self.thresh else 0
        return self.pmap

    def order_sensitive_cells(self):
        """Cells whose reference answer depends on the arrival order of +hit and +free (only
        possible when hit_inc <= thresh): h >= 1 hits below the last level and p exactly on
        the canonical threshold or one below it."""
        table = occupied_rule(self.free_inc, self.hit_inc, self.thresh)
        m = np.zeros(self.pass_cnt.shape, dtype=bool)
        for h, t in enumerate(table):
            if h >= 1:
                m |= (self.hit_cnt == h) & ((self.pass_cnt == t) | (self.pass_cnt == t - 1))
        return m

    def pmap_from_counts(self):
        """The integer rule the HIP finalize kernel applies: untouched -> 50;
        occupied per :func:`occupied_rule` -> 100 (for +20: hit >= 1 or pass >= 1001); else 0."""
        table = occupied_rule(self.free_inc, self.hit_inc, self.thresh)
        touched = (self.pass_cnt + self.hit_cnt) > 0
        occ = self.hit_cnt >= len(table)
        for h, t in enumerate(table):
            occ |= (self.hit_cnt == h) & (self.pass_cnt >= t)
        return np.where(touched, np.where(occ, 100, 0), 50).astype(np.int8)


def occupancy_grid_data(pmap):
    """W12m/slam_ekf.py:270-271: data[y*width + x] = int8(trunc(pmap[x][y]))."""
    return np.trunc(pmap.T.reshape(-1)).astype(np.int8)


# ----------------------------------------------------------------------------
# f-1  scan-to-map observation (W9 = "W9_Fusion Localization (LiDAR Odometry)/course_agv_slam/scripts")
# ----------------------------------------------------------------------------
def map_obstacles(data, width, height, resolution, origin_x, origin_y):
    """W9/localization.py:54-60 (updateMap): OccupancyGrid data (data[y*width + x]) ->
    obstacle coordinates [2, K].  Cells > 20 or < -0.5 count, i.e. occupied AND unknown."""
    map_data = np.array(data).reshape((-1, height)).transpose()
    tx, ty = np.nonzero((map_data > 20) | (map_data < -0.5))
    ox = (tx * resolution + origin_x) * 1.0
    oy = (ty * resolution + origin_y) * 1.0
    return np.vstack((ox, oy))


def laser_estimation(obstacle, x_est, angle_min, angle_increment, total_num):
    """W9/localization.py:128-150 (laserEstimation): the scan the map would produce from
    pose x_est = (x, y, theta): every obstacle point drops its distance into the beam bin
    int((atan2(dy, dx) - angle_min - theta) / angle_increment) (truncated, then wrapped into
    [0, total_num)), bins keep the minimum, empty bins stay at 100.0."""
    ranges = [100.0] * total_num
    for i in range(obstacle.shape[1]):
        dist = math.hypot(x_est[0] - obstacle[0][i], x_est[1] - obstacle[1][i])
        index = int((math.atan2(obstacle[1][i] - x_est[1], obstacle[0][i] - x_est[0]) - angle_min - x_est[2]) / angle_increment)
        while index > total_num - 1:
            index = index - total_num
        while index < 0:
            index = index + total_num
        if dist < ranges[index]:
            ranges[index] = dist
    return np.array(ranges)


def map_observation(obstacle, x_est, src_pc, angle_min, angle_max, angle_increment, max_iter=30, tolerance=0.001):
    """W9/localization.py:152-157 (calc_map_observation): ICP of the current scan against the
    virtual scan of the map; laserToNumpy of :168-174 on the float64 virtual ranges."""
    n = src_pc.shape[1]
    tar_pc = laser_to_numpy(laser_estimation(obstacle, x_est, angle_min, angle_increment, n), angle_min, angle_max)
    return icp_process(tar_pc, src_pc, max_iter, tolerance)


# ----------------------------------------------------------------------------
# pipeline (a-7 -> a-3 -> a-6 -> a-8 -> a-10), the unit bench.py counts as one scan
# ----------------------------------------------------------------------------
def replay(ranges, angle_min, angle_max, mapping, max_iter=30, tolerance=0.001,
           pose0=(0.0, 0.0, 0.0), nn=find_nearest):
    """Scan k (k >= 1) is matched against scan k-1 (slam_ekf.py:109-113), the pose is
    dead-reckoned with icp.py:185-190 (the EKF of slam_ekf.py:86 is out of scope,
    SURVEY.md a-8), and the scan is ray-cast into ``mapping`` from that pose
    (slam_ekf.py:89-90).  The first scan only becomes the target (:74-78).
    Returns poses [n_scan-1, 3], T [n_scan-1, 3, 3], iteration counts."""
    n_scan = ranges.shape[0]
    sta = [float(pose0[0]), float(pose0[1]), float(pose0[2])]
    poses = np.zeros((n_scan - 1, 3))
    ts = np.zeros((n_scan - 1, 3, 3))
    iters = np.zeros(n_scan - 1, dtype=np.int32)
    tar = laser_to_numpy(ranges[0], angle_min, angle_max, clip_inf=True)
    for k in range(1, n_scan):
        cur = laser_to_numpy(ranges[k], angle_min, angle_max, clip_inf=True)
        t, it, _ = icp_process(tar, cur, max_iter, tolerance, nn=nn, return_info=True)
        tar = cur
        sta = compose_pose(sta, t)
        obs = world_points(sta, cur)
        mapping.update(obs[0], obs[1], sta[0], sta[1])
        poses[k - 1] = sta
        ts[k - 1] = t
        iters[k - 1] = it
    return poses, ts, iters
