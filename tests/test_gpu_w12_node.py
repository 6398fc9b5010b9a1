"""The whole W12 node - SLAM_EKF(landmarks=True): device scan matching and map building around
the host-side landmark extraction and EKF - against the reference's own SLAM_EKF.laserCallback
run on the same 66-message stream (tests/golden/g7_w12_node.npz, recipe O5)."""
import numpy as np
import pytest

from conftest import load_golden, pkg

pytestmark = pytest.mark.gpu
AMIN, AMAX = -3.14159, 3.14159


@pytest.fixture(scope="module")
def slam():
    p = pkg()
    p._abi.default_context()
    return p


@pytest.fixture(scope="module")
def g7():
    return load_golden("g7_w12_node.npz")


def msg(slam, ranges):
    return slam.LaserScan(ranges=tuple(float(v) for v in ranges), angle_min=AMIN, angle_max=AMAX)


def test_observation_golden(slam, g7):
    node = slam.SLAM_EKF(landmarks=True)
    rows = []
    for k in range(len(g7["ext_offsets"]) - 1):
        lm = node.extraction.process(node.laserToNumpy(msg(slam, g7["ext_ranges"][k])))
        rows.append(node.observation(lm))
    assert np.max(np.abs(np.concatenate(rows) - g7["ext_z"])) < 1e-12


def test_full_node_golden(slam, g7):
    node = slam.SLAM_EKF(landmarks=True)
    steps = {int(k): i for i, k in enumerate(g7["node_steps"])}
    for k, r in enumerate(g7["node_ranges"]):
        before = node.xEst.copy()
        node.laserCallback(msg(slam, r))
        changed = node.xEst.shape != before.shape or not np.array_equal(node.xEst, before)
        assert changed == (k in steps), k
        if k in steps:
            i = steps[k]
            assert (len(node.xEst) - 3) // 2 == g7["node_nlm"][i], k
            assert np.max(np.abs(node.xEst[:3, 0] - g7["node_xest"][i])) < 1e-9, k
    assert np.max(np.abs(node.xEst[:, 0] - g7["node_final_x"])) < 1e-9
    assert np.max(np.abs(node.PEst - g7["node_final_P"])) < 1e-9
    assert np.array_equal(node.mapping.pmap.astype(np.int8), g7["node_pmap"])
    assert node.last_map["data"].shape == (200 * 200,)


def test_node_without_landmarks_does_nothing(slam, g7):
    """No landmark in view: the reference returns before odometry, filter and map (slam_ekf.py:80-82)."""
    node = slam.SLAM_EKF(landmarks=True)
    for k in range(12):
        node.laserCallback(msg(slam, g7["ext_empty_ranges"][k % 2]))
    assert np.array_equal(node.xEst, np.zeros((3, 1))) and np.all(node.mapping.pmap == 50) and node.extraction.flag == 1
