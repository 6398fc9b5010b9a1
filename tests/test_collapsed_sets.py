"""Collapsed correspondences: every source point matched to ONE target point (icp.py:149-179).

W = BB^T.AA is then mathematically zero and every rotation is optimal.  The reference's centred
target rows are rounding noise instead (np.mean of n equal values is not that value), W ~ 1e-31,
and the rotation its SVD returns is arbitrary - tests/golden/g8_collapsed.npz holds what the
reference itself returned on eight such cases (rotations by anything from 16 to 170 degrees).
DOCUMENTED DEVIATION: the oracles and the product return the canonical answer R = I (the SVD of
an exactly zero matrix), t = centroid_B - centroid_A.  Both answers move the source centroid
onto the matched point; only the (meaningless) rotation differs."""
import numpy as np
import pytest

from conftest import load_golden, pkg
from oracle import c_oracle as co
from oracle import oracle_np as on


@pytest.fixture(scope="module")
def g8():
    return load_golden("g8_collapsed.npz")


def canonical(src, tar_rows):
    T = np.eye(3)
    T[:2, 2] = tar_rows.mean(axis=1) - src.mean(axis=1)
    return T


def test_reference_answer_is_an_arbitrary_rotation_about_the_same_fixed_point(g8):
    for k in range(len(g8["src"])):
        src, rows, T = g8["src"][k], g8["tar_rows"][k], g8["T_ref"][k]
        assert np.all(rows == rows[:, :1])                                   # every target row is one point
        R = T[:2, :2]
        assert abs(np.linalg.det(R) - 1.0) < 1e-12 and np.max(np.abs(R.T @ R - np.eye(2))) < 1e-12
        assert np.max(np.abs(R @ src.mean(axis=1) + T[:2, 2] - rows[:, 0])) < 1e-9   # centroid -> the matched point
    angles = np.degrees(np.arctan2(g8["T_ref"][:, 1, 0], g8["T_ref"][:, 0, 0]))
    assert np.ptp(angles) > 90.0                                             # "arbitrary": all over the circle


def test_reference_faithful_mode_reproduces_the_reference(g8):
    """The NumPy restatement WITHOUT the canonical rule is the reference's own arithmetic: it returns what the
    reference returned (G8), so the restatement stays an independent check of W12m/icp.py:154-169 and the deviation
    is exactly the one rule.  Reference and canonical answer agree on where the source centroid goes."""
    for k in range(len(g8["src"])):
        src, rows, T_ref = g8["src"][k], g8["tar_rows"][k], g8["T_ref"][k]
        # (the memory layouts the reference saw - a transposed view and a fancy-indexed copy, oracle/gen_golden.py gen_g8:
        # np.mean sums a strided view in another order than a contiguous array, and here the answer IS that rounding)
        tar_rows = np.ascontiguousarray(rows.T)
        T = on.get_transform(src.T, tar_rows, collapsed_rule="reference")
        assert np.max(np.abs(T - T_ref)) < 1e-9, k
        Tc = on.get_transform(src.T, tar_rows)
        ca = src.mean(axis=1)
        assert np.max(np.abs((T[:2, :2] @ ca + T[:2, 2]) - (Tc[:2, :2] @ ca + Tc[:2, 2]))) < 1e-9
        assert np.max(np.abs(T[:2, :2] - Tc[:2, :2])) > 0.1                   # ... and differ in the rotation
    # on sets that are NOT collapsed the two modes are the same function
    rng = np.random.default_rng(5)
    a, b = rng.normal(0, 3, size=(40, 2)), rng.normal(0, 3, size=(40, 2))
    assert np.array_equal(on.get_transform(a, b), on.get_transform(a, b, collapsed_rule="reference"))


def test_oracles_return_the_canonical_answer(g8):
    for k in range(len(g8["src"])):
        src, rows = g8["src"][k], g8["tar_rows"][k]
        want = canonical(src, rows)
        for T in (on.get_transform(src.T, rows.T), on.get_transform_closed_form(src.T, rows.T), co.get_transform(src.T, rows.T)):
            assert np.array_equal(T[:2, :2], np.eye(2)) and np.max(np.abs(T - want)) < 1e-12
        # whole solves: after the first step every source point coincides... no: it is the same cloud
        # moved rigidly onto the point's neighbourhood; the solve is a fixed point of the canonical rule
        oT, oit, _ = co.icp_process(g8["cloud"][k], src, 30, 0.001)
        nT, nit, _ = on.icp_process(np.vstack([g8["cloud"][k], np.ones((1, 3))]), np.vstack([src, np.ones((1, src.shape[1]))]),
                                    return_info=True)
        assert oit == nit and np.max(np.abs(oT - nT)) < 1e-12
        assert np.max(np.abs(oT[:2, :2] - np.eye(2))) < 1e-12 and np.max(np.abs(oT - want)) < 1e-9   # (the final fit of the solve is an ordinary one)
    # a collapsed SOURCE set (all source points equal) is the same situation mirrored
    src = np.tile(np.array([[0.1], [0.7]]), (1, 9))
    tar = np.random.default_rng(0).normal(0, 2, size=(2, 9))
    T = co.get_transform(src.T, tar.T)
    assert np.array_equal(T[:2, :2], np.eye(2)) and np.max(np.abs(T - canonical(src, tar))) < 1e-12


@pytest.mark.gpu
def test_device_returns_the_canonical_answer(g8):
    slam = pkg()
    icp = slam.ICP()
    ones = lambda a: np.vstack([a, np.ones((1, a.shape[1]))])
    for k in range(len(g8["src"])):
        src, rows, cloud = g8["src"][k], g8["tar_rows"][k], g8["cloud"][k]
        want = canonical(src, rows)
        T = icp.getTransform(src.T, rows.T)                                  # the stand-alone operator (k_kabsch)
        assert np.array_equal(T[:2, :2], np.eye(2)) and np.max(np.abs(T - want)) < 1e-12
        T = icp.process(ones(cloud), ones(src))                              # the whole solve (k_icp, one query per lane)
        oT, oit, _ = co.icp_process(cloud, src, 30, 0.001)
        assert icp.last_iters == oit and np.max(np.abs(T - oT)) < 1e-9 and np.max(np.abs(T[:2, :2] - np.eye(2))) < 1e-12
    # batched form with more than one wave per pair and several queries per lane: 400 source points
    rng = np.random.default_rng(1)
    cloud = np.array([[0.3, 103.1, 211.7], [0.7, 97.3, -54.9]])
    srcs = np.stack([cloud[:, :1] + rng.normal(0, 0.2, size=(2, 400)) for _ in range(80)])
    T, it, _ = slam.icp_batch_host(np.broadcast_to(cloud, (80, 2, 3)).copy(), srcs, 30, 0.001)
    oT, oit, _ = co.icp_batch(np.broadcast_to(cloud, (80, 2, 3)).copy(), srcs, 30, 0.001)
    assert np.array_equal(it, oit) and np.max(np.abs(T - oT)) < 1e-9
    assert np.max(np.abs(T[:, 0, 0] - 1.0)) < 1e-12 and np.max(np.abs(T[:, 1, 0])) < 1e-12
    # ... in every launch shape of large batches (three queries per lane: the pair's first wave finishes the iteration
    # for the others - the collapsed-set test is then its business alone)
    for qpt in (1, 2, 3):
        ctx = slam.Context(0)
        ctx.set_option("icp_qpt", qpt)
        Tq, itq, _ = slam.icp_batch_host(np.broadcast_to(cloud, (80, 2, 3)).copy(), srcs, 30, 0.001, context=ctx)
        ctx.close()
        assert np.array_equal(itq, oit) and np.max(np.abs(Tq - oT)) < 1e-9, qpt
        assert np.max(np.abs(Tq[:, 0, 0] - 1.0)) < 1e-12 and np.max(np.abs(Tq[:, 1, 0])) < 1e-12
    # collapsed source set
    src = np.tile(np.array([[0.1], [0.7]]), (1, 9))
    tar = rng.normal(0, 2, size=(2, 9))
    T = icp.getTransform(src.T, tar.T)
    assert np.array_equal(T[:2, :2], np.eye(2)) and np.max(np.abs(T - canonical(src, tar))) < 1e-12
