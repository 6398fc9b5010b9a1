"""Host-side logic of the full W12 node against vectors produced by the
reference's own extraction.py and ekf_lm.py (tests/golden/g7_w12_node.npz, oracle/gen_golden.py
recipe O5).  No GPU needed: these two classes are NumPy on the host in the product as well."""
import numpy as np
import pytest

from conftest import load_golden, pkg
from oracle import oracle_np as on

AMIN, AMAX = -3.14159, 3.14159


@pytest.fixture(scope="module")
def g7():
    return load_golden("g7_w12_node.npz")


def cloud(ranges):
    return on.laser_to_numpy(np.asarray(ranges), AMIN, AMAX, clip_inf=True)


def test_extraction_golden(g7):
    ex = pkg("extraction").Extraction()
    offs = g7["ext_offsets"]
    for k in range(len(offs) - 1):
        lm = ex.process(cloud(g7["ext_ranges"][k]))
        a, b = offs[k], offs[k + 1]
        assert lm is not None and lm.id == g7["ext_id"][a:b].tolist(), k
        assert np.array_equal(np.array(lm.position_x), g7["ext_x"][a:b]), k      # same sums in the same order: exact
        assert np.array_equal(np.array(lm.position_y), g7["ext_y"][a:b]), k
    assert ex.flag == 0
    for k in range(2):                                                           # an empty room has no landmark
        assert ex.process(cloud(g7["ext_empty_ranges"][k])) is None and g7["ext_empty_is_none"][k]
    assert ex.flag == 1


def test_extraction_label_rules():
    """The labelling rules spelled out in extraction.py's docstring, on hand-made point rows."""
    ex = pkg("extraction").Extraction()
    x = np.array([0.0, 0.05, 0.10, 5.0, 9.0, 9.05, 20.0, 20.05, 20.10, 20.15])
    pc = np.vstack([x, np.zeros_like(x), np.ones_like(x)])
    labels, found = ex.labels(pc)
    # cluster 0 = points 0..2 closed by the gap after point 2; point 3 is alone (-1); points 4,5:
    # only ONE earlier member when the gap comes -> point 5 is -1 and point 4 keeps number 2;
    # the last cluster (3) is never closed
    assert labels.tolist() == [0, 0, 0, -1, 2, -1, 3, 3, 3] and found == [0]
    lm = ex.process(pc)
    assert lm.id == [0] and lm.position_x == [(0.0 + 0.05 + 0.10) / 3] and lm.position_y == [0.0]
    wide = pc.copy()
    wide[0, 2] = 0.4                                    # extent 0.4 >= radius_max_th: a cluster, not a landmark
    assert ex.process(wide) is None
    assert ex.process(pc[:, :1]) is None and ex.process(pc[:, :0]) is None


def test_ekf_sequence_golden(g7):
    ekf = pkg("ekf_lm").EKF()
    xE, PE = np.zeros((3, 1)), np.eye(3)
    zo = g7["ekf_z_offsets"]
    for t in range(len(g7["ekf_sizes"])):
        z = g7["ekf_z"][zo[t]:zo[t + 1]]
        xE, PE = ekf.estimate(xE, PE, z, g7["ekf_u"][t].reshape(3, 1))
        n = int(g7["ekf_sizes"][t])
        assert len(xE) == n and PE.shape == (n, n), t
        assert np.max(np.abs(xE[:, 0] - g7["ekf_x"][t][:n])) < 1e-9, t
        assert np.max(np.abs(PE - g7["ekf_P"][t][:n, :n])) < 1e-9, t
    assert int(g7["ekf_sizes"][-1]) == 11        # four landmarks found, one per call at most


def test_ekf_quirks():
    m = pkg("ekf_lm")
    ekf = m.EKF()
    assert ekf.pi_2_pi(3 * np.pi / 2) == pytest.approx(-np.pi / 2) and m.STATE_SIZE == 3 and m.M_DIST_TH == 0.6
    # two unknown landmarks in one call: the first is added, the second finds no slot and ends the
    # call early, BEFORE the yaw normalisation (ekf_lm.py:40-42)
    x, P = np.array([[0.0], [0.0], [3.0]]), np.eye(3)
    z = np.array([[2.0, 0.1, 0], [3.0, -1.0, 1]])
    x2, P2 = ekf.estimate(x, P, z, np.array([[0.1], [0.0], [0.5]]))
    assert len(x2) == 5 and P2.shape == (5, 5) and x2[2, 0] > np.pi
    # the prediction writes into the caller's arrays, like the reference
    assert x[2, 0] == 3.5
    # a known landmark is matched (no growth) and the yaw is wrapped at the end
    x3, P3 = ekf.estimate(x2.copy(), P2.copy(), np.array([[2.0, 0.1, 0]]), np.zeros((3, 1)))
    assert len(x3) == 5 and -np.pi <= x3[2, 0] < np.pi


def test_w9_pose_filter_golden():
    """localization.EKF (the 3-state filter of the W9 node) against the reference's own ekf.py."""
    g5 = load_golden("g5_map_observation.npz")
    ekf = pkg("localization").EKF()
    x, P = np.array([0.1, -0.2, 0.05]), np.eye(3)
    for k in range(g5["ekf9_x"].shape[0]):
        x, P = ekf.estimate(x, P, g5["ekf9_z"][k], g5["ekf9_T"][k])
        assert np.max(np.abs(np.asarray(x, dtype=float) - g5["ekf9_x"][k])) < 1e-12, k
        assert np.max(np.abs(P - g5["ekf9_P"][k])) < 1e-12, k
