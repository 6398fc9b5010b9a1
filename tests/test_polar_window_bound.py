"""The index window of the beam-window nearest-neighbour search (csrc/icp_kernels.hip nn_polar), restated
in NumPy float32 and checked against the exhaustive search on the CPU: for a query s and a guess beam j at
squared distance U, every target at most that far lies on a beam k with

    j + ceil(min(0, (dlo - alpha) * inv_db))  <=  k  <=  j + floor(max(0, (dhi + alpha) * inv_db))

(alpha >= asin(sqrt(U) / |s|), [dlo, dhi] bracketing the angle between s and the guess's ray, inv_db >= 1 /
smallest beam spacing; every factor rounded outwards by >= 1e-6).  The kernel adds nothing to this window
(it used to add one beam either side); the GPU parity tests check the kernel, this test checks the formula -
including a reciprocal and a square root that are one unit in the last place off, as the hardware's are."""
import numpy as np
import pytest

f32 = np.float32


def window(qx, qy, tx, ty, seed, inv_db, slack_ang, rcp_err):
    U = (qx - tx[seed]) ** 2 + (qy - ty[seed]) ** 2
    fsx, fsy, ftx, fty = qx.astype(f32), qy.astype(f32), tx[seed].astype(f32), ty[seed].astype(f32)
    rs2 = fsx * fsx + fsy * fsy
    x2 = (U.astype(f32) * f32(1.000002) + f32(1e-30)) * (f32(1) / (rs2 * f32(0.999998))) * f32(1 - rcp_err)
    small = x2 < f32(0.25)
    x = np.sqrt(x2) * f32(1 - rcp_err) * f32(1.000001)
    alpha = x * (f32(1) + f32(0.19) * x2) * f32(1.000002) + f32(slack_ang)
    y = (ftx * fsy - fty * fsx) * (f32(1) / (ftx * fsx + fty * fsy))
    y = y * np.where(y >= 0, f32(1 + rcp_err), f32(1 - rcp_err)).astype(f32) if rcp_err else y
    y3 = y * y * y * f32(0.33333334)
    dhi = np.where(y >= 0, y, y - y3) + f32(4e-6)
    dlo = np.where(y >= 0, y - y3, y) - f32(4e-6)
    lo = seed + np.ceil(np.minimum(f32(0), (dlo - alpha) * inv_db)).astype(np.int64)
    hi = seed + np.floor(np.maximum(f32(0), (dhi + alpha) * inv_db)).astype(np.int64)
    return small, lo, hi


@pytest.mark.parametrize("n,span", [(360, 4.712), (360, 3.0), (1080, 4.712), (90, 3.0), (22, 4.0), (500, 1.0)])
@pytest.mark.parametrize("kind", ["noise", "steps", "smooth", "circle"])
def test_window_holds_every_target_as_close_as_the_guess(n, span, kind):
    rng = np.random.default_rng(n * 7 + len(kind))
    ang = np.linspace(-span / 2, span / 2, n)
    ct, st = np.cos(ang), np.sin(ang)
    cr = ct[:-1] * st[1:] - st[:-1] * ct[1:]
    inv_db = f32(f32(1.000002) / f32(f32(cr.min()) * f32(0.999999)))
    checked = 0
    for trial in range(40 if n < 1000 else 12):
        if kind == "noise":
            r = rng.uniform(0.1, 20, n)
        elif kind == "steps":
            r = np.round(rng.uniform(0.5, 8) + np.cumsum(rng.integers(-1, 2, n)) * 0.25, 2).clip(0.25, 30)
        elif kind == "circle":
            r = np.full(n, rng.uniform(0.5, 10))                     # every beam equally far: the window's edges are hit exactly
        else:
            r = 5 + np.sin(ang * 3 + rng.uniform(0, 6)) * 2 + rng.normal(0, 0.01, n)
        r = r.astype(np.float32).astype(np.float64)
        tx, ty = ct * r, st * r
        th, tr = rng.normal(0, 0.05), rng.normal(0, 0.1, 2)
        r2 = (r * (1 + rng.normal(0, 0.02, n))).clip(0.05, 40)
        qx = np.cos(th) * ct * r2 - np.sin(th) * st * r2 + tr[0]
        qy = np.sin(th) * ct * r2 + np.cos(th) * st * r2 + tr[1]
        seed = np.clip(np.arange(n) + rng.integers(-3, 4, n), 0, n - 1)
        d2 = (qx[:, None] - tx[None, :]) ** 2 + (qy[:, None] - ty[None, :]) ** 2
        U = d2[np.arange(n), seed]
        for rcp_err in (0.0, 1.2e-7):
            small, lo, hi = window(qx, qy, tx, ty, seed, inv_db, 2e-7, rcp_err)
            k = np.arange(n)[None, :]
            outside = (k < lo[:, None]) | (k > hi[:, None])
            as_close = d2 <= U[:, None] * (1 + 2.0 ** -49)           # the guess's distance, and the class of equal roots above it
            bad = small[:, None] & outside & as_close
            assert not bad.any(), (n, span, kind, trial, np.argwhere(bad)[:5])
        checked += int(small.sum())
    assert checked > 100


def _perturbed_pair(rng, ct, st, ang, n, kind):
    if kind == "steps":
        r = np.round(rng.uniform(0.5, 8) + np.cumsum(rng.integers(-1, 2, n)) * 0.25, 2).clip(0.25, 30)
    else:
        r = 5 + np.sin(ang * 3 + rng.uniform(0, 6)) * 2 + rng.normal(0, 0.01, n)
    r = r.astype(np.float32).astype(np.float64)
    tx, ty = ct * r, st * r
    th, tr = rng.normal(0, 0.02), rng.normal(0, 0.03, 2)
    r2 = (r * (1 + rng.normal(0, 0.005, n))).clip(0.05, 40)
    qx = np.cos(th) * ct * r2 - np.sin(th) * st * r2 + tr[0]
    qy = np.sin(th) * ct * r2 + np.cos(th) * st * r2 + tr[1]
    return tx, ty, qx, qy


@pytest.mark.parametrize("n,span", [(360, 4.712), (1080, 4.712), (90, 3.0)])
def test_window_with_a_near_neighbour_as_the_guess(n, span):
    """The regime of the iterations after the first: the guess is the previous match, i.e. (one of) the nearest
    targets, the bound is tight and the window a beam or two - where rounding the index bounds inwards (floor above,
    ceil below) instead of outwards matters: every target as close as the guess must still be inside."""
    rng = np.random.default_rng(n)
    ang = np.linspace(-span / 2, span / 2, n)
    ct, st = np.cos(ang), np.sin(ang)
    cr = ct[:-1] * st[1:] - st[:-1] * ct[1:]
    inv_db = f32(f32(1.000002) / f32(f32(cr.min()) * f32(0.999999)))
    checked, widths = 0, []
    for trial in range(24 if n < 1000 else 8):
        tx, ty, qx, qy = _perturbed_pair(rng, ct, st, ang, n, "steps" if trial % 3 == 0 else "smooth")
        d2 = (qx[:, None] - tx[None, :]) ** 2 + (qy[:, None] - ty[None, :]) ** 2
        order = np.argsort(d2, axis=1, kind="stable")
        for rank in range(4):
            seed = order[:, rank]
            U = d2[np.arange(n), seed]
            for rcp_err in (0.0, 1.2e-7):
                small, lo, hi = window(qx, qy, tx, ty, seed, inv_db, 2e-7, rcp_err)
                k = np.arange(n)[None, :]
                outside = (k < lo[:, None]) | (k > hi[:, None])
                as_close = d2 <= U[:, None] * (1 + 2.0 ** -49)
                bad = small[:, None] & outside & as_close
                assert not bad.any(), (n, span, trial, rank, np.argwhere(bad)[:5])
            checked += int(small.sum())
            if rank == 0:
                widths.append((hi - lo + 1)[small])
    assert checked > 1000
    assert np.median(np.concatenate(widths)) <= max(3, n // 90)      # (the tight regime is what is being tested)


def test_the_series_bound_of_asin():
    """alpha uses x (1 + 0.19 x^2) >= asin(x) on [0, 0.5] ((asin x - x) / x^3 grows from 1/6 to 0.18879 at 0.5)."""
    x = np.linspace(0.0, 0.5, 200001)
    assert np.all(x * (1 + 0.19 * x * x) >= np.arcsin(x))
    xf = x.astype(f32)
    assert np.all((xf * (f32(1) + f32(0.19) * xf * xf) * f32(1.000002)).astype(np.float64) >= np.arcsin(xf.astype(np.float64)))


def kernel_ranges(lo, hi, n):
    """The three ascending index ranges nn_polar scans for a window [lo, hi] on a scan that may close on itself
    (csrc/icp_kernels.hip): wrapped from above [0, e0] | the window [m0, m1] | wrapped from below [s2, n - 1]."""
    m0, m1 = np.maximum(lo, 0), np.minimum(hi, n - 1)
    e0 = np.where(hi >= n - 2, np.minimum(hi - (n - 1) + 1, m0 - 1), -1)
    s2 = np.where(lo <= 1, np.maximum(n - 1 + lo - 1, m1 + 1), n)
    return e0, m0, m1, s2


@pytest.mark.parametrize("gap", [5.3e-6, 0.0, -0.45, 0.6, 1.0])
@pytest.mark.parametrize("n", [360, 90])
def test_wrapped_ranges_of_a_scan_that_closes_on_itself(n, gap):
    """Full-circle scans: beam n - 1 and beam 0 are neighbours `gap` beam spacings apart (the benchmark's linspace(-3.14159,
    3.14159) leaves 5.3e-6 rad; polar_probe admits a last beam up to half a spacing PAST the first).  Every target as close
    as the guess lies in one of the three index ranges the kernel scans."""
    rng = np.random.default_rng(n + int(gap * 100))
    if gap == 5.3e-6:
        ang = np.linspace(-3.14159, 3.14159, n)
    else:
        ang = -np.pi + 2 * np.pi / (n - 1 + gap) * np.arange(n)
    ct, st = np.cos(ang), np.sin(ang)
    cr = ct[:-1] * st[1:] - st[:-1] * ct[1:]
    inv_db = f32(f32(1.000002) / f32(f32(cr.min()) * f32(0.999999)))
    checked = wrapped = 0
    for trial in range(30):
        tx, ty, qx, qy = _perturbed_pair(rng, ct, st, ang, n, "steps" if trial % 3 == 0 else "smooth")
        if trial % 2:                                  # a larger turn: matches several beams away, across the seam too
            th = rng.normal(0, 0.08)
            qx, qy = np.cos(th) * qx - np.sin(th) * qy, np.sin(th) * qx + np.cos(th) * qy
        d2 = (qx[:, None] - tx[None, :]) ** 2 + (qy[:, None] - ty[None, :]) ** 2
        order = np.argsort(d2, axis=1, kind="stable")
        for rank in range(3):
            seed = order[:, rank]
            U = d2[np.arange(n), seed]
            small, lo, hi = window(qx, qy, tx, ty, seed, inv_db, 2e-7, 1.2e-7)
            e0, m0, m1, s2 = kernel_ranges(lo, hi, n)
            k = np.arange(n)[None, :]
            inside = (k <= e0[:, None]) | ((k >= m0[:, None]) & (k <= m1[:, None])) | (k >= s2[:, None])
            as_close = d2 <= U[:, None] * (1 + 2.0 ** -49)
            bad = small[:, None] & ~inside & as_close
            assert not bad.any(), (n, gap, trial, rank, np.argwhere(bad)[:5])
            # the three ranges never overlap (a target is compared once, in ascending index order)
            assert np.all(e0 < m0) and np.all(s2 > m1)
            checked += int(small.sum())
            wrapped += int((small & ((e0 >= 0) | (s2 < n))).sum())
    assert checked > 1000 and wrapped > 50
