"""Nearest-neighbour ORDERING on sub-ulp near-ties: what the reference, the oracle and the
kernel each pick, pinned so that the difference is documented instead of hidden.

The reference orders candidates by ``np.linalg.norm(src[i] - tar[j])`` with strict '<' (W12m
icp.py:102-103): sqrt(x.dot(x)).  The 2-element dot is BLAS arithmetic; in this container's
NumPy / OpenBLAS it is fma(dy, dy, dx*dx) (checked below), on another build it may be the unfused
dx*dx + dy*dy.  sqrt maps about half of all pairs of adjacent doubles to one value, so two
candidates whose squared distances differ by one unit in the last place are a TIE for the
reference (the lower index wins) while they are ordered for anything that compares squares.

  kernel (nn_search / nn_polar / k_nn):  argmin of d2 = fma(dy, dy, dx*dx), lowest index among equal d2
  C / NumPy oracle:                      argmin of sqrt(dx*dx + dy*dy) (unfused), lowest index among equals
  reference here:                        argmin of sqrt(fma(dy, dy, dx*dx)), lowest index among equals

They agree unless two candidates' squared distances differ by <= 2 ulp (probability ~1e-16 per
comparison on real scans; exact ties - equal coordinates, quantised clouds - are unaffected:
every rule then picks the lowest index).  The cases below are constructed to sit on that edge."""
import math
from fractions import Fraction

import numpy as np
import pytest

from conftest import pkg
from oracle import c_oracle as co


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))       # correctly rounded


def d2_fused(dx, dy):
    return fma(dy, dy, dx * dx)


def rule_kernel(src, tar):
    d = [d2_fused(src[0] - t[0], src[1] - t[1]) for t in tar]
    return int(np.argmin(d))                                      # first minimum


def rule_oracle(src, tar):
    d = [math.sqrt((src[0] - t[0]) * (src[0] - t[0]) + (src[1] - t[1]) * (src[1] - t[1])) for t in tar]
    return int(np.argmin(d))


def rule_reference_here(src, tar):
    best, idx = float("inf"), 0                                   # the reference's loop, icp.py:99-105
    for j, t in enumerate(tar):
        dist = np.linalg.norm(np.asarray(src) - np.asarray(t))
        if dist < best:
            best, idx = dist, j
    return idx


def sqrt_collapse_case():
    """Two targets whose fused squared distances are adjacent doubles with ONE square root; the
    lower index has the larger square."""
    rng = np.random.default_rng(3)
    while True:
        dx, dy = rng.uniform(0.5, 2.0, 2)
        a, dx2 = d2_fused(dx, dy), dx
        for _ in range(5):
            dx2 = np.nextafter(dx2, 0.0)
            b = d2_fused(dx2, dy)
            if b == np.nextafter(a, 0.0) and math.sqrt(a) == math.sqrt(b):
                return (0.0, 0.0), [(float(dx), float(dy)), (float(dx2), float(dy))]


def fused_order_case():
    """Two targets whose UNFUSED squares are equal while the fused ones differ; the lower index
    has the larger fused square."""
    rng = np.random.default_rng(4)
    while True:
        dx, dy = rng.uniform(0.5, 2.0, 2)
        dx2 = dx
        for _ in range(3):
            dx2 = np.nextafter(dx2, 3.0)
            if d2_fused(dx, dy) < d2_fused(dx2, dy) and dx * dx + dy * dy == dx2 * dx2 + dy * dy:
                return (0.0, 0.0), [(float(dx2), float(dy)), (float(dx), float(dy))]


CASES = {"sqrt_collapse": sqrt_collapse_case, "fused_vs_unfused": fused_order_case}


def test_numpy_dot_of_two_elements_is_fused_here():
    """Premise of the table above: this container's x.dot(x) on 2 elements is fma(x1, x1, x0*x0)."""
    rng = np.random.default_rng(0)
    seen = 0
    for _ in range(20000):
        a, b = rng.normal(size=2)
        unf, fus = a * a + b * b, fma(b, b, a * a)
        if unf != fus:
            seen += 1
            assert float(np.array([a, b]).dot(np.array([a, b]))) == fus
    assert seen > 1000


@pytest.mark.parametrize("name", sorted(CASES))
def test_near_tie_rules_cpu(name):
    src, tar = CASES[name]()
    k, o, r = rule_kernel(src, tar), rule_oracle(src, tar), rule_reference_here(src, tar)
    d, i = co.find_nearest(np.array([src]), np.array(tar))
    assert int(i[0]) == o                                         # the C oracle follows its stated rule
    if name == "sqrt_collapse":
        # one square root for both: a tie for the reference and the oracle (index 0); the squares
        # are ordered, so anything comparing squares picks index 1
        assert (k, o, r) == (1, 0, 0)
    else:
        # equal unfused squares: a tie for the oracle (index 0); the fused squares differ by one ulp:
        # the kernel picks index 1, the reference picks 1 unless their square roots collapse
        assert (k, o) == (1, 0) and r in (0, 1)
    # and with a third, clearly nearer, candidate every rule agrees again
    tar3 = tar + [(tar[0][0] * 0.5, tar[0][1] * 0.5)]
    assert rule_kernel(src, tar3) == rule_oracle(src, tar3) == rule_reference_here(src, tar3) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_near_tie_rules_gpu(name):
    """The device follows ITS stated rule (argmin of the fused square, lowest index among equal
    squares) in all three searches: the stand-alone operator (k_nn), and - through ICP.process -
    the box search; exact ties still go to the lowest index."""
    slam = pkg()
    src, tar = CASES[name]()
    icp = slam.ICP()
    S, Tg = np.array([src]), np.array(tar)
    d, i = icp.findNearest(S, Tg)
    assert int(i[0]) == rule_kernel(src, tar) == 1
    assert d[0] == math.sqrt(d2_fused(src[0] - tar[1][0], src[1] - tar[1][1]))
    # exact tie (the same point twice): lowest index, as every rule says
    d, i = icp.findNearest(S, np.array([tar[1], tar[1], tar[0]]))
    assert int(i[0]) == 0
    # padded to a block and more: the near-tied pair in the middle of a larger cloud
    far = [(50.0 + k, 60.0) for k in range(40)]
    cloud = np.array(far[:20] + tar + far[20:])
    d, i = icp.findNearest(S, cloud)
    assert int(i[0]) == 21
