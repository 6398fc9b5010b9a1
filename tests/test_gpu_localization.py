"""GPU parity of the scan-to-map observation against the golden vectors made
by the reference's own Localization (tests/golden/g5_map_observation.npz) and the CPU oracle.

Bars: obstacle list bit-exact (as a set: the device appends in arbitrary order, the Python
mirror restores np.nonzero order); beam bins of the virtual scan identical, its distances
within 1e-12 (device hypot vs CPython's math.hypot differ in the last bit at most);
transforms within 1e-9."""
import types

import numpy as np
import pytest

from conftest import load_golden, pkg
from oracle import oracle_np as on

pytestmark = pytest.mark.gpu
AMIN, AMAX = -3.14159, 3.14159


@pytest.fixture(scope="module")
def slam():
    p = pkg()
    p._abi.default_context()
    return p


@pytest.fixture(scope="module")
def g5():
    return load_golden("g5_map_observation.npz")


def grid_msg(data, w=200, h=200, res=0.1, ox=-10.0, oy=-10.0):
    return types.SimpleNamespace(data=data, info=types.SimpleNamespace(
        height=h, width=w, resolution=res, origin=types.SimpleNamespace(position=types.SimpleNamespace(x=ox, y=oy))))


def scan_msg(slam, ranges, n):
    return slam.LaserScan(ranges=tuple(float(v) for v in ranges), angle_min=AMIN, angle_max=AMAX,
                          angle_increment=(AMAX - AMIN) / (n - 1))


def test_update_map_golden(slam, g5):
    loc = slam.Localization()
    loc.updateMap(grid_msg(g5["map_data"]))
    assert loc.obstacle.shape == g5["obstacle"].shape
    assert np.array_equal(loc.obstacle, g5["obstacle"])            # same values AND np.nonzero order
    assert loc.obstacle_r == 0.1


def test_update_map_layouts_and_edges(slam):
    rng = np.random.default_rng(3)
    loc = slam.Localization()
    for (w, h) in ((1, 1), (7, 7), (64, 64), (33, 33)):
        data = rng.choice(np.array([-1, 0, 20, 21, 50, 100], dtype=np.int8), size=w * h)
        loc.updateMap(grid_msg(data, w, h, 0.05, -1.25, 2.5))
        want = on.map_obstacles(data, w, h, 0.05, -1.25, 2.5)
        assert np.array_equal(loc.obstacle, want)
    loc.updateMap(grid_msg(np.zeros(100, dtype=np.int8), 10, 10))     # no obstacle at all
    assert loc.obstacle.shape == (2, 0)
    with pytest.raises(ValueError):
        loc.updateMap(grid_msg(np.zeros(10, dtype=np.int8), 3, 3))    # numpy's reshape error in the reference


@pytest.mark.parametrize("n,key", [(120, "vscan120"), (360, "vscan360")])
def test_laser_estimation_golden(slam, g5, n, key):
    loc = slam.Localization()
    loc.obstacle = g5["obstacle"]
    msg = scan_msg(slam, [1.0] * n, n)
    for k, pose in enumerate(g5["poses"]):
        loc.xEst = [float(v) for v in pose]
        est = loc.laserEstimation(msg, loc.xEst)
        got = np.array(est.ranges)
        assert len(est.ranges) == n and est.angle_min == msg.angle_min
        assert np.array_equal(got == 100.0, g5[key][k] == 100.0), (n, k)       # same bins filled
        assert np.max(np.abs(got - g5[key][k])) < 1e-12, (n, k)
    # batched: all hypotheses in one launch == one at a time
    batch = loc.virtual_ranges(msg, g5["poses"])
    assert np.max(np.abs(batch - g5[key])) < 1e-12
    # heading comes from the argument, position from xEst (localization.py:138-139)
    loc.xEst = [float(v) for v in g5["poses"][0]]
    other = loc.laserEstimation(msg, [99.0, -99.0, float(g5["poses"][1][2])])
    want = on.laser_estimation(g5["obstacle"], [g5["poses"][0][0], g5["poses"][0][1], g5["poses"][1][2]], AMIN,
                               msg.angle_increment, n)
    assert np.max(np.abs(np.array(other.ranges) - want)) < 1e-12


def test_laser_estimation_edges(slam):
    loc = slam.Localization()
    msg = scan_msg(slam, [1.0] * 16, 16)
    loc.xEst = [0.0, 0.0, 0.0]
    assert loc.laserEstimation(msg, loc.xEst).ranges == [100.0] * 16          # no obstacles yet
    loc.obstacle = np.array([[1.0, 2.0, -1.0, 200.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0]])
    for th in (0.0, 7.0, -7.0, 40.0):                                         # wrap in both directions, many turns
        got = np.array(loc.laserEstimation(msg, [0.0, 0.0, th]).ranges)
        want = on.laser_estimation(loc.obstacle, [0.0, 0.0, th], AMIN, msg.angle_increment, 16)
        assert np.array_equal(got, want), th                                   # exact: distances are integers here
    # an obstacle on the robot: distance 0 lands in atan2(0, 0) = 0's bin
    got = np.array(loc.laserEstimation(msg, [0.0, 0.0, 0.0]).ranges)
    assert got.min() == 0.0


def test_laser_to_numpy_f64(slam, g5):
    loc = slam.Localization()
    r = g5["vscan120"][0]
    pc = loc.laserToNumpy(scan_msg(slam, r, 120))
    assert np.array_equal(pc, on.laser_to_numpy(r, AMIN, AMAX))
    r32 = np.float32(r)
    pc = loc.laserToNumpy(scan_msg(slam, r32, 120))
    assert np.array_equal(pc, on.laser_to_numpy(r32.astype(np.float64), AMIN, AMAX))


def test_map_observation_golden(slam, g5):
    loc = slam.Localization()
    loc.obstacle = g5["obs_wall"]
    for k in range(g5["obs_T"].shape[0]):
        msg = scan_msg(slam, g5["obs_ranges"][k], 120)
        loc.xEst = [float(v) for v in g5["obs_xest"][k]]
        loc.src_pc = loc.laserToNumpy(msg)
        T = loc.calc_map_observation(msg)
        assert np.max(np.abs(T - g5["obs_T"][k])) < 1e-9, k


def test_map_observation_batch_vs_oracle(slam, g5, syn):
    """Many pose hypotheses of one scan in one call == the oracle one at a time."""
    rng = np.random.default_rng(8)
    loc = slam.Localization()
    loc.obstacle = g5["obs_wall"]
    n = 360
    true_pose = np.array([0.7, -0.4, 0.3])
    r = syn.scans_from_poses(syn.World(5.0, 4.0, (), 0.0), true_pose[None], n, 5)[0]
    msg = scan_msg(slam, r, n)
    loc.src_pc = loc.laserToNumpy(msg)
    poses = true_pose + rng.normal(0, [0.1, 0.1, 0.03], size=(24, 3))
    T, it = loc.map_observation_batch(msg, poses)
    src = on.laser_to_numpy(np.asarray(msg.ranges), AMIN, AMAX)
    for k in range(poses.shape[0]):
        want = on.map_observation(g5["obs_wall"], poses[k], src, AMIN, AMAX, msg.angle_increment)
        assert np.max(np.abs(T[k] - want)) < 1e-9, k
    assert it.min() >= 1


def test_localization_callback_sequence(slam, g5, syn):
    """laserCallback over a short replay against the same steps restated with the oracle."""
    loc = slam.Localization()
    loc.obstacle = g5["obs_wall"]
    world = syn.World(5.0, 4.0, (), 0.0)
    poses = np.array([[0.02 * k, 0.01 * k, 0.004 * k] for k in range(18)])
    scans = syn.scans_from_poses(world, poses, 120, 9)
    ekf = slam.localization.EKF()
    xest, pest, xodom = [0, 0, 0], np.eye(3), [0, 0, 0]
    tar = None
    inc = (AMAX - AMIN) / 119
    comp = slam.localization._compose
    for k in range(scans.shape[0]):
        msg = scan_msg(slam, scans[k], 120)
        loc.laserCallback(msg)
        if (k + 1) % 6:
            continue
        src = on.laser_to_numpy(np.asarray(msg.ranges), AMIN, AMAX)
        if tar is None:
            tar = on.laser_to_numpy(on.laser_estimation(g5["obs_wall"], xest, AMIN, inc, 120), AMIN, AMAX)
        T = on.icp_process(tar, src, 30, 0.001)
        xodom = comp(xodom, T)
        T = on.icp_process(src, src, 30, 0.001)
        t = on.map_observation(g5["obs_wall"], xest, src, AMIN, AMAX, inc)
        new2 = comp(xest, t)
        xest, pest = ekf.estimate(xest, pest, new2, T)
        tar = src
        assert np.max(np.abs(np.asarray(loc.xEst) - xest)) < 1e-9, k
        assert np.max(np.abs(np.asarray(loc.xOdom) - xodom)) < 1e-9, k
        assert np.max(np.abs(loc.PEst - pest)) < 1e-9


def test_w9_node_golden(slam, g5):
    """Localization.laserCallback on the reference's own run of the whole W9 node (real ekf.py):
    6 processed scans, each = odometry ICP (twice, as the reference does) + map observation +
    pose filter."""
    loc = slam.Localization()
    loc.obstacle = g5["obs_wall"]
    steps = {int(k): i for i, k in enumerate(g5["node9_steps"])}
    for k, r in enumerate(g5["node9_ranges"]):
        loc.laserCallback(scan_msg(slam, r, 120))
        if k in steps:
            i = steps[k]
            assert np.max(np.abs(np.asarray(loc.xEst, dtype=float) - g5["node9_xest"][i])) < 1e-9, k
            assert np.max(np.abs(np.asarray(loc.xOdom, dtype=float) - g5["node9_xodom"][i])) < 1e-9, k
    assert np.max(np.abs(loc.PEst - g5["node9_P"])) < 1e-9 and len(steps) == 6
