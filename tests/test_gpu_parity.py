"""GPU parity tests: the HIP path, called through the C ABI (libslamhip.so), against
(1) the golden vectors produced by the reference's own code and (2) the CPU oracle on the
same seeded inputs.  Run on the MI355X box with ``-m gpu``.

Bars (BASELINE.json north_star): cell indices / counters / iteration counts / NN indices
bit-exact; float64 poses and transforms within 1e-9 (target bar: 1e-5 m / 1e-5 rad)."""
import zlib

import numpy as np
import pytest

from conftest import pkg
from oracle import c_oracle as co
from oracle import oracle_np as on

pytestmark = pytest.mark.gpu
FTOL = 1e-9
AMIN, AMAX = -3.14159, 3.14159


@pytest.fixture(scope="module")
def slam():
    p = pkg()
    p._abi.default_context()   # fails loudly when there is no gfx950 device
    return p


class Msg:
    def __init__(self, ranges, angle_min=AMIN, angle_max=AMAX):
        self.ranges = tuple(float(v) for v in ranges)
        self.angle_min, self.angle_max = angle_min, angle_max


# ------------------------------------------------------------------ bresenham (G1)
def test_bresenham_fan_golden(slam, g1):
    ends = g1["fan_ends"].astype(np.int32)
    paths = slam.rasterize(np.zeros_like(ends), ends)
    cells, offs = g1["fan_cells"], g1["fan_offsets"]
    for k, p in enumerate(paths):
        assert np.array_equal(p, cells[offs[k]:offs[k + 1]].astype(np.int32)), ends[k]


def test_bresenham_random_golden(slam, g1):
    paths = slam.rasterize(g1["rand_starts"], g1["rand_ends"])
    for k, p in enumerate(paths):
        assert len(p) == g1["rand_len"][k]
        assert zlib.crc32(p.astype("<i4").tobytes()) == g1["rand_crc"][k], k


def test_bresenham_class_surface(slam):
    b = slam.bresenham([3, 4], [3, 4])
    assert b.path == [] and b.flag == 0
    b = slam.bresenham([0, 0], [-7, 3])
    assert b.path == on.bresenham_path([0, 0], [-7, 3]) and b.flag == 1 and b.path[0] == (0, 0) and b.path[-1] == (-7, 3)
    b = slam.bresenham((2, 9), (4, -20))
    assert b.path == on.bresenham_path([2, 9], [4, -20]) and b.steep


# ------------------------------------------------------------------ Mapping (G2)
@pytest.mark.parametrize("n", [120, 200, 360])
def test_mapping_demo_golden(slam, g2, n):
    m = slam.Mapping(200, 200, 0.1)
    for i in range(10):
        c = g2["demo%d_c" % n][i]
        p = m.update(g2["demo%d_ox" % n][i], g2["demo%d_oy" % n][i], c[0], c[1])
        assert p is m.pmap and p.dtype == np.float64
        assert np.array_equal(p.astype(np.int8), g2["demo%d_pmap_steps" % n][i]), i
    assert np.max(np.abs(m.datamap - g2["demo%d_datamap" % n])) < 1e-9


def test_mapping_static_centre_golden(slam, g2):
    m = slam.Mapping(200, 200, 0.1)
    c = g2["static_c"]
    for i in range(5):
        p = m.update(g2["static_ox"][i], g2["static_oy"][i], np.array([c[0]]), np.array([c[1]]))
        assert np.array_equal(p.astype(np.int8), g2["static_pmap_steps"][i]), i
    assert np.max(np.abs(m.datamap - g2["static_datamap"])) < 1e-9


@pytest.mark.parametrize("case,xw,yw", [("edge", 200, 200), ("outside", 200, 200), ("rect", 120, 260)])
def test_mapping_cases_golden(slam, g2, case, xw, yw):
    m = slam.Mapping(xw, yw, 0.1)
    c = g2[case + "_c"]
    p = m.update(g2[case + "_ox"], g2[case + "_oy"], c[0], c[1])
    assert np.array_equal(p.astype(np.int8), g2[case + "_pmap"])
    assert np.max(np.abs(m.datamap - g2[case + "_datamap"])) < 1e-9
    og = co.Grid(xw, yw)
    og.update(g2[case + "_ox"], g2[case + "_oy"], c[0], c[1])
    ps, ht = m.counters()
    assert np.array_equal(ps, og.pass_cnt) and np.array_equal(ht, og.hit_cnt)
    assert m.visits() == og.visits
    assert np.array_equal(m.occupancy_grid_data(), og.occupancy_grid_data())


def test_mapping_errors_like_int(slam):
    m = slam.Mapping(200, 200, 0.1)
    with pytest.raises(ValueError):
        m.update(np.array([1.0, np.nan]), np.array([0.0, 0.0]), 0.0, 0.0)
    with pytest.raises(OverflowError):
        m.update(np.array([1.0]), np.array([np.inf]), 0.0, 0.0)
    with pytest.raises(ValueError):
        m.update(np.array([1.0]), np.array([1.0]), np.nan, 0.0)
    m = slam.Mapping(200, 200, 0.1)
    p = m.update(np.array([np.inf, np.inf]), np.array([np.nan, 1.0]), np.nan, 0.0)   # every beam skipped: no error
    assert np.all(p == 50)
    with pytest.raises(OverflowError):
        m.update(np.array([1.0e9]), np.array([0.0]), 0.0, 0.0)                     # beyond the 2^20-cell ray limit
    p = m.update(np.array([0.5]), np.array([0.0]), 0.0, 0.0)                       # still usable afterwards
    assert p[105, 100] == 100 and p[100, 100] == 0 and np.sum(p != 50) == 6


def test_mapping_pass_threshold_1001(slam):
    """The 1001st pass flips a never-hit cell to occupied (sequential float64 sum)."""
    g = slam.DeviceGrid(1, 200, 200, 10.0, 10.0, 10.0)
    ox = np.full((250, 4), 0.55)
    for rep in range(1, 5):
        g.update_host(ox, np.zeros_like(ox), np.zeros(250), np.zeros(250))
        r = g.read(0, want=("pmap", "pass", "hit"))
        assert r["pass"][101, 100] == 1000 * rep and r["hit"][105, 100] == 1000 * rep
        assert r["pmap"][101, 100] == (0 if rep == 1 else 100)
    g.update_host(ox[:1, :1] * 0 + 0.15, np.zeros((1, 1)), np.zeros(1), np.zeros(1))
    assert g.read(0)["pmap"][100, 100] == 100


def test_generalised_index_rule_vs_oracle(slam, syn):
    """400x400 @ 0.05 (scale 20, offset 10) and 2000x2000 @ 0.02 (50, 20): cfg2 / cfg5 grids."""
    rng = np.random.default_rng(5)
    for (xw, reso, rmax) in ((400, 0.05, 9.0), (2000, 0.02, 22.0)):
        g = slam.DeviceGrid.metric(1, xw, xw, reso)
        s = 1.0 / reso
        og = co.Grid(xw, xw, round(s), xw / (2 * round(s)), xw / (2 * round(s)))
        for _ in range(3):
            ang = np.linspace(-np.pi, np.pi, 360)
            d = rng.random(360) * rmax + 0.3
            c = rng.uniform(-2, 2, size=2)
            ox, oy = c[0] + np.cos(ang) * d, c[1] + np.sin(ang) * d
            g.update_host(ox, oy, c[0], c[1])
            og.update(ox, oy, c[0], c[1])
        r = g.read(0, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt)
        assert np.array_equal(r["pmap"], og.pmap) and g.visits() == og.visits


# ------------------------------------------------------------------ ICP (G3)
def test_laser_to_numpy_golden(slam, g3):
    icp = slam.ICP()
    for k in range(int(g3["nn_count"])):
        r = g3["nn%d_ranges" % k]
        pc = slam.scan_to_pc(Msg(r[1]), clip_inf=True)
        assert np.array_equal(pc, g3["nn%d_src" % k])
        assert np.array_equal(icp.laserToNumpy(Msg(r[0])), g3["nn%d_tar" % k])   # no inf in these scans
    r = np.array([1.0, np.inf, 2.0], dtype=np.float32)
    assert np.hypot(*slam.scan_to_pc(Msg(r), clip_inf=True)[:2, 1]) == 30.0
    assert not np.isfinite(icp.laserToNumpy(Msg(r))[0, 1])


def test_find_nearest_golden(slam, g3):
    icp = slam.ICP()
    for k in range(int(g3["nn_count"])):
        tar, src = g3["nn%d_tar" % k], g3["nn%d_src" % k]
        d, i = icp.findNearest(src[:2, :].transpose(), tar[:2, :].transpose())
        assert i.dtype == np.int64 and np.array_equal(i, g3["nn%d_idx" % k])
        assert np.max(np.abs(d - g3["nn%d_dist" % k])) < 1e-14


def test_find_nearest_ties_nan_and_sizes(slam):
    icp = slam.ICP()
    tar = np.array([[1.0, 0.0], [-1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    src = np.array([[0.0, 0.0], [np.nan, 0.0], [1.0, 0.0]])
    d, i = icp.findNearest(src, tar)
    assert list(i) == [0, 0, 0] and list(d) == [1.0, 0.0, 0.0]
    rng = np.random.default_rng(1)
    for n, m in ((1, 1), (5, 700), (1500, 3), (777, 1081)):
        s, t = rng.normal(size=(n, 2)), rng.normal(size=(m, 2))
        d, i = icp.findNearest(s, t)
        dd, ii = co.find_nearest(s, t)
        assert np.array_equal(i, ii) and np.max(np.abs(d - dd)) < 1e-14


def test_get_transform_golden(slam, g3):
    icp = slam.ICP()
    for k in range(100):
        a, b = g3["gt_src"][k], g3["gt_tar"][k]
        n = int(np.sum(~np.isnan(a[:, 0])))
        T = icp.getTransform(a[:n], b[:n])
        assert np.max(np.abs(T - g3["gt_T"][k])) < FTOL, (k, bool(g3["gt_reflect"][k]))


def test_icp_process_golden(slam, g3):
    icp = slam.ICP()
    for k in range(g3["pr_T"].shape[0]):
        n = int(g3["pr_n"][k])
        mi, tol = int(g3["pr_cfg"][k, 0]), float(g3["pr_cfg"][k, 1])
        tar = slam.scan_to_pc(Msg(g3["pr_ranges"][k, 0, :n]), clip_inf=True)
        src = slam.scan_to_pc(Msg(g3["pr_ranges"][k, 1, :n]), clip_inf=True)
        slam.param.set_param('/icp/tolerance', tol)
        icp.max_iter = mi
        try:
            T = icp.process(tar, src)
        finally:
            slam.param.clear_params()
        assert icp.last_iters == g3["pr_iters"][k], k
        assert np.max(np.abs(T - g3["pr_T"][k])) < FTOL, k
        assert abs(icp.last_mean_error - g3["pr_mean_err"][k]) < FTOL


def test_icp_ragged_golden(slam, g3):
    icp = slam.ICP()
    tar = slam.scan_to_pc(Msg(g3["rag_r0"]), clip_inf=True)
    src = slam.scan_to_pc(Msg(g3["rag_r1"]), clip_inf=True)
    T = icp.process(tar, src)
    assert icp.last_iters == int(g3["rag_iters"]) and np.max(np.abs(T - g3["rag_T"])) < FTOL


@pytest.mark.parametrize("n,m", [(1, 1), (2, 3), (63, 64), (65, 129), (1080, 1080), (1500, 900), (2500, 700), (5000, 64), (300, 4545), (700, 8192), (8192, 8192)])
def test_icp_batch_sizes_vs_oracle(slam, n, m):
    """Edge sizes incl. more than one query per lane (n > 1024), tiny clouds and targets up to the documented maximum
    of 8 192 points (the kernel's LDS then holds the padded copy of the target alone: ADVICE r2)."""
    rng = np.random.default_rng(n * 7 + m)
    B = 3
    tar = rng.normal(0, 3, size=(B, 2, m))
    th = rng.uniform(-0.1, 0.1, size=B)
    src = np.empty((B, 2, n))
    for b in range(B):
        pick = rng.integers(0, m, size=n)
        c, s = np.cos(th[b]), np.sin(th[b])
        x, y = tar[b, 0, pick] + rng.normal(0, 0.02, n), tar[b, 1, pick] + rng.normal(0, 0.02, n)
        src[b, 0], src[b, 1] = c * x - s * y + 0.05, s * x + c * y - 0.03
    T, it, err = slam.icp_batch_host(tar, src, 12, 1e-4)
    oT, oit, oerr = co.icp_batch(tar, src, 12, 1e-4)
    assert np.array_equal(it, oit)
    assert np.max(np.abs(T - oT)) < FTOL and np.max(np.abs(err - oerr)) < FTOL


@pytest.mark.parametrize("beams", [4544, 4545, 8192])
def test_replay_largest_scans_vs_oracle(slam, syn, beams):
    """Scans of up to 8 192 beams through the fused replay: up to 4 544 beams the scan matcher keeps a second, unpadded
    copy of the target in LDS for its beam-window search, beyond that both copies no longer fit and it goes by the box
    search alone - same answers either way (iteration counts exact, poses to 1e-9)."""
    rep = syn.make_replay(3, beams, seed=21, stride=5)
    ctx = slam.Context(0)
    poses, T, it = slam.replay_host(rep.ranges, AMIN, AMAX, context=ctx)
    oposes, oT, oit, _ = co.replay(rep.ranges, AMIN, AMAX, None, threads=8)
    assert np.array_equal(it, oit)
    assert np.max(np.abs(poses - oposes)) < FTOL and np.max(np.abs(T - oT.reshape(T.shape))) < FTOL
    ctx.close()


@pytest.mark.parametrize("gap", [5.3e-6, 0.0, -0.45, 0.6, 1.0])
@pytest.mark.parametrize("n", [360, 90])
def test_replay_scans_that_close_on_themselves(slam, syn, n, gap):
    """Full-circle scans whose last beam lies `gap` beam spacings before the first again (the benchmark's linspace(-3.14159,
    3.14159) leaves 5.3e-6 rad; a scan may even overlap itself by up to half a spacing): the beam-window search scans the
    wrapped index ranges on both sides of the seam (csrc/icp_kernels.hip nn_polar; the rule is checked on the CPU by
    tests/test_polar_window_bound.py::test_wrapped_ranges_of_a_scan_that_closes_on_itself).  Every other scan sees other
    surfaces in the beams next to the seam, so that windows there are wide and cross it.  Against the oracle's exhaustive
    search: iteration counts exact, poses and transforms to 1e-9."""
    rng = np.random.default_rng(int(n + 100 * gap))
    ranges = syn.make_replay(20, n, seed=31, stride=5).ranges.copy()
    ranges[1::2, :5] *= rng.uniform(0.6, 1.5, size=ranges[1::2, :5].shape).astype(np.float32)
    ranges[1::2, -5:] *= rng.uniform(0.6, 1.5, size=ranges[1::2, -5:].shape).astype(np.float32)
    amin = -np.pi
    amax = AMAX if gap == 5.3e-6 else amin + 2 * np.pi * (n - 1) / (n - 1 + gap)
    if gap == 5.3e-6:
        amin = AMIN
    ctx = slam.Context(0)
    poses, T, it = slam.replay_host(ranges, amin, amax, context=ctx)
    ctx.close()
    oposes, oT, oit, _ = co.replay(ranges, amin, amax, None, threads=8)
    assert np.array_equal(it, oit)
    assert np.max(np.abs(poses - oposes)) < FTOL and np.max(np.abs(T - oT.reshape(T.shape))) < FTOL


@pytest.mark.parametrize("kind", ["replay", "jumps", "random", "dense", "five"])
def test_listed_first_iteration_queries_match_the_box_search(slam, syn, kind):
    """First-iteration queries without a usable beam window are listed in LDS and searched apart from the lanes that own
    them (context option icp_team 0: re-guess + window per listed query, exhaustive rows of 16 lanes for what is
    left), or by the box search of their own lane (icp_team 1): bit-identical T and iteration counts, and both equal
    the oracle.  'jumps': every other beam of the source scans at an unrelated range (half the queries listed: the list
    overflows and the rest takes the box search); 'random': no surface at all (every listed query ends in the
    exhaustive rows); 'dense': 1 080 beams; 'five': a five-beam scan (rows mostly empty)."""
    rng = np.random.default_rng(77)
    if kind == "dense":
        ranges = syn.make_replay(6, 1080, seed=23, stride=5, room_scale=2.0).ranges
    elif kind == "five":
        ranges = rng.uniform(0.5, 6.0, size=(9, 5)).astype(np.float32)
    else:
        ranges = syn.make_replay(24, 360, seed=22, stride=5).ranges.copy()
        if kind == "jumps":
            ranges[1::2, ::2] = rng.uniform(0.5, 20.0, size=ranges[1::2, ::2].shape).astype(np.float32)
        if kind == "random":
            ranges = rng.uniform(0.3, 25.0, size=ranges.shape).astype(np.float32)
    out = []
    for team in (0, 1):
        ctx = slam.Context(0)
        ctx.set_option("icp_team", team)
        out.append(slam.replay_host(ranges, AMIN, AMAX, context=ctx))
        ctx.close()
    (p0, T0, it0), (p1, T1, it1) = out
    assert np.array_equal(it0, it1) and np.array_equal(T0, T1) and np.array_equal(p0, p1)
    oposes, oT, oit, _ = co.replay(ranges, AMIN, AMAX, None, threads=8)
    assert np.array_equal(it0, oit)
    assert np.max(np.abs(p0 - oposes)) < FTOL and np.max(np.abs(T0 - oT.reshape(T0.shape))) < FTOL


def test_icp_max_iter_zero_and_tol_zero(slam, syn):
    pair = syn.scan_pair(120, seed=1)
    tar, src = co.laser_to_points(pair.ranges[0], AMIN, AMAX), co.laser_to_points(pair.ranges[1], AMIN, AMAX)
    tar, src = np.array(tar)[None], np.array(src)[None]
    for mi, tol in ((0, 0.001), (1, 0.0), (10, 0.0)):
        T, it, _ = slam.icp_batch_host(tar, src, mi, tol)
        oT, oit, _ = co.icp_batch(tar, src, mi, tol)
        assert it[0] == oit[0] == mi and np.max(np.abs(T - oT)) < FTOL


@pytest.mark.parametrize("dtype,npdt", [("f32", np.float32), ("f16", np.float16)])
def test_icp_reduced_storage_matches_oracle_on_rounded_points(slam, syn, dtype, npdt):
    """f32 / f16 are STORAGE types of the point buffers; arithmetic stays float64, so the
    result must equal the oracle fed the same rounded points."""
    rep = syn.make_replay(9, 360, seed=12)
    pts = np.stack([np.array(co.laser_to_points(r, AMIN, AMAX)) for r in rep.ranges]).astype(npdt)
    tar, src = pts[:-1], pts[1:]
    T, it, err = slam.icp_batch_host(tar, src, 30, 0.001, dtype=dtype)
    oT, oit, oerr = co.icp_batch(tar.astype(np.float64), src.astype(np.float64), 30, 0.001)
    assert np.array_equal(it, oit) and np.max(np.abs(T - oT)) < FTOL
    # and the device conversion rounds exactly like NumPy's
    dev = np.empty((2, 360), dtype=npdt)
    ct, st = slam._abi.trig_tables(AMIN, AMAX, 360)
    r0 = np.ascontiguousarray(rep.ranges[0])
    slam._abi.check(slam._abi.lib().slam_scan_to_points(slam.default_context().handle, r0.ctypes.data, ct.ctypes.data,
                                                       st.ctypes.data, 1, 360, 1, slam._abi.DTYPES[dtype], dev.ctypes.data))
    assert np.array_equal(dev, pts[0])


@pytest.mark.parametrize("dtype,npdt", [("f32", np.float32), ("f16", np.float16)])
def test_replay_reduced_storage_vs_oracle(slam, syn, dtype, npdt):
    """The fused replay with f32 / f16 point storage (BASELINE configs[1] says fp32, configs[4]
    fp16): scan matching sees the points rounded to that type (formed in registers, never
    stored), the map is still cast from float64 world points."""
    rep = syn.make_replay(25, 360, seed=14, stride=5)
    grid = slam.DeviceGrid.metric(1, 400, 400, 0.05)
    poses, T, it = slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid, dtype=dtype)
    pts64 = np.stack([np.array(co.laser_to_points(r, AMIN, AMAX)) for r in rep.ranges])
    pts = pts64.astype(npdt).astype(np.float64)
    oT, oit, _ = co.icp_batch(pts[:-1], pts[1:], 30, 0.001)
    assert np.array_equal(it, oit) and np.max(np.abs(T - oT.reshape(T.shape))) < FTOL
    og = co.Grid(400, 400, 20.0, 10.0, 10.0)
    sta = [0.0, 0.0, 0.0]
    for k in range(24):
        sta = co.compose_pose(sta, oT[k].reshape(3, 3))
        assert np.max(np.abs(poses[k] - np.array(sta))) < FTOL
        wx, wy = co.world_points(poses[k], pts64[k + 1][0], pts64[k + 1][1])     # device poses: same cells as the device
        og.update(wx, wy, poses[k][0], poses[k][1])
    r = grid.read(0, want=("pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt)
    # and the storage type does change the answer (otherwise this test would prove nothing)
    _, T64, _ = slam.replay_host(rep.ranges, AMIN, AMAX)
    assert np.max(np.abs(T64 - T)) > 1e-9


def test_particle_priors_vs_oracle(slam, syn):
    """cfg3 operator: one scan pair, P perturbed priors (shared target and source)."""
    pair = syn.scan_pair(360, seed=2)
    tar = np.array(co.laser_to_points(pair.ranges[0], AMIN, AMAX))
    src = np.array(co.laser_to_points(pair.ranges[1], AMIN, AMAX))
    pri = syn.particle_priors(64, seed=2)
    mats = np.zeros((64, 2, 3))
    mats[:, 0, 0], mats[:, 0, 1], mats[:, 0, 2] = np.cos(pri[:, 2]), -np.sin(pri[:, 2]), pri[:, 0]
    mats[:, 1, 0], mats[:, 1, 1], mats[:, 1, 2] = np.sin(pri[:, 2]), np.cos(pri[:, 2]), pri[:, 1]
    T, it, err = slam.icp_batch_host(tar, src, 30, 0.001, prior=mats)
    srcs = np.stack([np.stack([m[0, 0] * src[0] + m[0, 1] * src[1] + m[0, 2], m[1, 0] * src[0] + m[1, 1] * src[1] + m[1, 2]])
                     for m in mats])
    oT, oit, oerr = co.icp_batch(np.repeat(tar[None], 64, 0), srcs, 30, 0.001)
    assert np.array_equal(it, oit) and np.max(np.abs(T - oT)) < FTOL
    assert len(set(it.tolist())) > 1   # the batch really holds different solves


def test_pose_compose_vs_oracle(slam):
    rng = np.random.default_rng(3)
    L, n = 3, 2500     # longer than one 1024-step chunk
    th = rng.uniform(-0.1, 0.1, size=(L, n))
    T = np.zeros((L, n, 9))
    T[..., 0], T[..., 1], T[..., 2] = np.cos(th), -np.sin(th), rng.normal(0, 0.05, (L, n))
    T[..., 3], T[..., 4], T[..., 5] = np.sin(th), np.cos(th), rng.normal(0, 0.05, (L, n))
    T[..., 8] = 1
    p0 = rng.normal(size=(L, 3))
    out = np.empty((L, n, 3))
    A = slam._abi
    A.check(A.lib().slam_pose_compose(slam.default_context().handle, T.ctypes.data, p0.ctypes.data, L, n, out.ctypes.data))
    for l in range(L):
        s = p0[l].copy()
        for k in range(n):
            s = co.compose_pose(s, T[l, k].reshape(3, 3))
            assert np.max(np.abs(out[l, k] - s)) < 1e-10, (l, k)


# ------------------------------------------------------------------ pipeline (G4)
@pytest.mark.parametrize("tag", ["a", "b"])
def test_pipeline_golden_fused(slam, g4, tag):
    ranges = g4[tag + "_ranges"]
    grid = slam.DeviceGrid(1, 200, 200, 10.0, 10.0, 10.0)
    poses, T, it = slam.replay_host(ranges, AMIN, AMAX, grid=grid)
    assert np.max(np.abs(poses - g4[tag + "_poses"])) < FTOL
    assert np.max(np.abs(T - g4[tag + "_T"])) < FTOL
    r = grid.read(0, want=("pmap", "datamap"))
    assert np.array_equal(r["pmap"], g4[tag + "_pmap"])
    assert np.max(np.abs(r["datamap"] - g4[tag + "_datamap"])) < 1e-9
    assert np.array_equal(grid.occupancy_grid_data(0), g4[tag + "_grid_data"])


def test_pipeline_golden_node_callbacks(slam, g4):
    """The same stream fed message by message through the SLAM_EKF shim (decimation: every
    5th message is processed, slam_ekf.py:65-68)."""
    ranges = g4["a_ranges"]
    node = slam.SLAM_EKF()
    k_proc = 0
    for k in range(ranges.shape[0]):
        for rep in range(5):   # 4 dropped + 1 processed
            node.laserCallback(Msg(ranges[k]))
        if k >= 1:
            assert np.max(np.abs(node.xEst[:3, 0] - g4["a_poses"][k - 1])) < FTOL, k
            k_proc += 1
    assert k_proc == ranges.shape[0] - 1
    assert np.array_equal(node.mapping.pmap.astype(np.int8), g4["a_pmap"])
    assert np.array_equal(node.last_map["data"], g4["a_grid_data"])
    assert node.last_map["width"] == 200 and node.last_map["origin"][:2] == (-10.0, -10.0)


def test_w7_node_callbacks_vs_oracle(slam, syn):
    """W7 ICP.laserCallback: first scan = target, then every 6th message (icp.py:51-54),
    launch-file parameters max_iter=10 / tolerance=0 (W7/launch/icp.launch:10-11)."""
    rep = syn.make_replay(31, 120, seed=8)
    slam.param.set_param('/icp/max_iter', 10)
    slam.param.set_param('/icp/tolerance', 0)
    try:
        icp = slam.ICP()
    finally:
        slam.param.clear_params()
    sta = [0.0, 0.0, 0.0]
    prev = on.laser_to_numpy(rep.ranges[0], AMIN, AMAX)
    for k in range(31):
        icp.laserCallback(Msg(rep.ranges[k]))
        if k > 0 and k % 6 == 0:
            cur = on.laser_to_numpy(rep.ranges[k], AMIN, AMAX)
            T, it, _ = co.icp_process(prev, cur, 10, 0.0)
            sta = list(co.compose_pose(sta, T))
            prev = cur
            assert icp.last_iters == it == 10
            assert np.max(np.abs(np.array(icp.sensor_sta) - np.array(sta))) < FTOL, k
    assert icp.last_odom["position"][2] == 0.001 and icp.last_odom["frame_id"] == "world_base"


# ------------------------------------------------------------------ replay vs oracle at larger sizes
def test_replay_200_scans_vs_oracle(slam, syn):
    rep = syn.make_replay(200, 360, seed=21, stride=5)
    grid = slam.DeviceGrid.metric(1, 400, 400, 0.05)
    poses, T, it = slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid)
    og = co.Grid(400, 400, 20.0, 10.0, 10.0)
    oposes, oT, oit, visits = co.replay(rep.ranges, AMIN, AMAX, og, threads=8)
    assert np.array_equal(it, oit)
    assert np.max(np.abs(poses - oposes)) < FTOL and np.max(np.abs(T - oT)) < FTOL
    r = grid.read(0, want=("pmap", "pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt)
    assert np.array_equal(r["pmap"], og.pmap) and grid.visits() == visits


def test_multi_trajectory_replay_and_grid_routing(slam, syn):
    """L = 3 streams into 2 maps (streams 0 and 2 share map 1): trajectories stay
    independent, maps add up."""
    reps = [syn.make_replay(30, 120, seed=30 + l) for l in range(3)]
    ranges = np.stack([r.ranges for r in reps])
    grid = slam.DeviceGrid(2, 200, 200, 10.0, 10.0, 10.0)
    p0 = np.array([[0, 0, 0], [1.0, -1.0, 0.5], [-2.0, 0.5, -1.0]])
    poses, T, it = slam.replay_host(ranges, AMIN, AMAX, grid=grid, pose0=p0, grid_of_traj=[1, 0, 1])
    ogs = [co.Grid(200, 200), co.Grid(200, 200)]
    for l, gi in enumerate([1, 0, 1]):
        op, oT, oit, _ = co.replay(ranges[l], AMIN, AMAX, ogs[gi], pose0=p0[l])
        assert np.array_equal(it[l], oit) and np.max(np.abs(poses[l] - op)) < FTOL
    for gi in range(2):
        r = grid.read(gi, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], ogs[gi].pass_cnt) and np.array_equal(r["hit"], ogs[gi].hit_cnt)
        assert np.array_equal(r["pmap"], ogs[gi].pmap)


def test_dense_1080_beam_config(slam, syn):
    """cfg5 shape: 1080 beams (two queries per lane), 2000x2000 @ 0.02 m map, room x2."""
    rep = syn.make_replay(6, 1080, seed=3, room_scale=2.0)
    grid = slam.DeviceGrid.metric(1, 2000, 2000, 0.02)
    poses, T, it = slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid)
    og = co.Grid(2000, 2000, 50.0, 20.0, 20.0)
    oposes, oT, oit, visits = co.replay(rep.ranges, AMIN, AMAX, og, threads=8)
    assert np.array_equal(it, oit) and np.max(np.abs(poses - oposes)) < FTOL
    r = grid.read(0, want=("pmap", "pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and grid.visits() == visits


# ------------------------------------------------------------------ full-size properties (BASELINE configs[1])
def test_full_size_replay(slam, syn):
    """BASELINE configs[1] at full size: 1k processed scans (every 5th message of a 10 Hz
    stream), 360 beams, 400x400 @ 0.05 m, buffers resident in HBM (slam_replay_dev).
    Compared in full with the C oracle, plus size-independent properties: run-to-run bit
    identity, exact doubling of the counters under a second accumulation, visit count ==
    sum of counters."""
    import torch
    rep = syn.make_replay(1000, 360, seed=1, stride=5)
    dr = slam.DeviceReplay(rep.ranges, AMIN, AMAX)
    grid = dr.make_grid(1, 400, 400, 0.05)
    dr.run()
    poses1, T1, it1 = dr.results()
    c1 = grid.read(0, want=("pmap", "pass", "hit"))
    v1 = grid.visits()
    og = co.Grid(400, 400, 20.0, 10.0, 10.0)
    oposes, oT, oit, ov = co.replay(rep.ranges, AMIN, AMAX, og, threads=8)
    assert np.array_equal(it1[0], oit)
    assert np.max(np.abs(poses1[0] - oposes)) < FTOL and np.max(np.abs(T1[0] - oT)) < FTOL
    assert np.array_equal(c1["pass"], og.pass_cnt) and np.array_equal(c1["hit"], og.hit_cnt)
    assert np.array_equal(c1["pmap"], og.pmap) and v1 == ov
    dr.run()                       # reset + rerun: identical bits
    poses2, T2, it2 = dr.results()
    c2 = grid.read(0, want=("pmap", "pass", "hit"))
    assert np.array_equal(poses1, poses2) and np.array_equal(T1, T2) and np.array_equal(it1, it2)
    assert all(np.array_equal(c1[k], c2[k]) for k in c1) and grid.visits() == v1
    dr.run(reset_grid=False)       # accumulate a second pass: counters double exactly
    c3 = grid.read(0, want=("pass", "hit"))
    assert np.array_equal(c3["pass"], 2 * c1["pass"]) and np.array_equal(c3["hit"], 2 * c1["hit"])
    assert int(c1["pass"].sum()) + int(c1["hit"].sum()) == v1
    assert set(np.unique(c1["pmap"]).tolist()) <= {0, 50, 100}
    assert np.array_equal(c1["pmap"] == 50, (c1["pass"] + c1["hit"]) == 0)
    assert it1.min() >= 2 and it1.max() <= 30
    torch.cuda.synchronize()


def test_live_pmap_matches_finalize(slam, syn):
    """slam_grid_live_pmap: ray casts that own their map keep pmap current cell for cell (per-
    particle maps, single-scan Mapping.update); shared-map updates mark it stale and the next
    read refreshes it.  Always equal to the full finalize of the counters."""
    rep = syn.make_replay(2, 360, seed=2, stride=5)
    P = 24
    priors = slam.prior_matrices(syn.particle_priors(P, seed=2))
    for live in (False, True):
        grid = slam.DeviceGrid.metric(P, 400, 400, 0.05)
        if live:
            assert grid.live_pmap() != 0
        for _ in range(3):                                   # maps persist: three filter steps
            slam.particles_host(rep.ranges[0], rep.ranges[1], AMIN, AMAX, priors, np.zeros((P, 3)), grid=grid)
        maps = [grid.read(g, want=("pmap", "pass", "hit")) for g in range(P)]
        if not live:
            want = maps
    for a, b in zip(maps, want):
        assert np.array_equal(a["pass"], b["pass"]) and np.array_equal(a["hit"], b["hit"]) and np.array_equal(a["pmap"], b["pmap"])
    assert any(np.any(m["pmap"] == 100) for m in maps) and any(np.any(m["pmap"] == 0) for m in maps)
    # single map: one scan per call is exclusive (live), a 40-scan replay is not (stale -> refreshed)
    rep = syn.make_replay(40, 360, seed=6, stride=5)
    outs = []
    for live in (False, True):
        grid = slam.DeviceGrid.metric(1, 400, 400, 0.05)
        if live:
            grid.live_pmap()
        slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid)
        r1 = grid.read(0, want=("pmap",))["pmap"].copy()
        ox = np.linspace(-3, 3, 50)[None]
        grid.update_host(ox, 0.3 * ox + 1.0, np.array([0.2]), np.array([-0.1]))
        r2 = grid.read(0, want=("pmap",))["pmap"].copy()
        grid.reset()
        r3 = grid.read(0, want=("pmap",))["pmap"].copy()
        outs.append((r1, r2, r3))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert np.all(outs[1][2] == 50) and not np.array_equal(outs[1][0], outs[1][1])


def test_counters_alias_checkpoint_and_merge(slam, syn):
    """slam_grid_counters_dev: the torch views alias the device counters - checkpoint / restore,
    and the merge of maps built from disjoint scans (what all_reduce(SUM) does across ranks) is
    the map built from all of them."""
    import torch
    rep = syn.make_replay(41, 360, seed=5, stride=5)
    whole = slam.DeviceGrid.metric(1, 400, 400, 0.05)
    poses, _, _ = slam.replay_host(rep.ranges, AMIN, AMAX, grid=whole)
    want = whole.read(0, want=("pmap", "pass", "hit"))
    # two "ranks": scans 1..20 and 21..40 cast from the same poses into two maps, then summed in place
    parts = []
    for lo, hi in ((0, 20), (20, 40)):
        g = slam.DeviceGrid.metric(1, 400, 400, 0.05)
        g.live_pmap()
        m_ranges = np.ascontiguousarray(rep.ranges[1 + lo:1 + hi])
        ct, st = slam._abi.trig_tables(AMIN, AMAX, 360)
        A = slam._abi
        A.check(A.lib().slam_grid_update_scans(g._ctx.handle, g._h, A.ptr(m_ranges), A.ptr(ct), A.ptr(st),
                                               A.ptr(np.ascontiguousarray(poses[lo:hi])), None, hi - lo, 360))
        parts.append(g)
    p0, h0 = parts[0].counters_torch()
    p1, h1 = parts[1].counters_torch()
    assert p0.dtype == torch.int32 and tuple(p0.shape) == (1, 400, 400) and p0.is_cuda
    snap = (p0.clone(), h0.clone())                           # checkpoint of part 0
    p0 += p1
    h0 += h1
    torch.cuda.synchronize()
    got = parts[0].read(0, want=("pmap", "pass", "hit"))      # the live pmap was marked stale -> refreshed
    assert np.array_equal(got["pass"], want["pass"]) and np.array_equal(got["hit"], want["hit"])
    assert np.array_equal(got["pmap"], want["pmap"])
    p0.copy_(snap[0]); h0.copy_(snap[1])                      # restore
    torch.cuda.synchronize()
    p0b, _ = parts[0].counters_torch()
    back = parts[0].read(0, want=("pass", "hit", "pmap"))
    assert int(back["pass"].astype(np.int64).sum() + back["hit"].astype(np.int64).sum()) < int(want["pass"].astype(np.int64).sum())
    assert np.array_equal(back["pass"], snap[0].cpu().numpy()[0].astype(np.uint32)) and p0b.data_ptr() == p0.data_ptr()
    slam.dist.all_reduce_grid(parts[0])                       # no process group: a no-op


def test_pipelined_map_stage_is_bit_identical(slam, syn):
    """"pipeline" option: the map stage on a second stream (overlapping the next replay's scan
    matching) gives the same bits as the serial order, with alternating AND with reused pose
    buffers, and with main-stream map calls interleaved."""
    import torch
    reps = [syn.make_replay(120, 360, seed=s, stride=5) for s in (3, 4)]
    want = []
    for rep in reps:                                         # serial reference
        dr = slam.DeviceReplay(rep.ranges, AMIN, AMAX)
        grid = dr.make_grid(1, 400, 400, 0.05)
        dr.run()
        want.append((dr.results(), grid.read(0, want=("pmap", "pass", "hit")), grid.visits()))
    A = slam._abi
    drs = [slam.DeviceReplay(rep.ranges, AMIN, AMAX) for rep in reps]
    ctx = drs[0].ctx
    drs[1].ctx = ctx                                         # both replays on ONE context
    ctx.set_option("pipeline", 1)
    grid = drs[0].make_grid(1, 400, 400, 0.05)
    drs[1].grid = grid
    pmaps = [torch.empty((400, 400), dtype=torch.int8, device=drs[0].dev) for _ in range(6)]
    bufs = [torch.empty_like(drs[0].poses) for _ in range(2)]
    tbufs = [torch.empty_like(drs[0].T) for _ in range(2)]
    for k in range(6):                                       # replays 3,4,3,4,3,4 back to back, no host sync
        d = drs[k & 1]
        d.run(reset_grid=True, poses_out=bufs[k & 1] if k < 4 else bufs[0],  # last two reuse ONE poses buffer,
              T_out=tbufs[k & 1] if k < 2 else tbufs[0])                       # from the third on ONE T buffer
        A.check(A.lib().slam_grid_finalize_dev(ctx.handle, grid._h, pmaps[k].data_ptr()))
    ctx.synchronize()
    torch.cuda.synchronize()
    for k in range(6):
        assert np.array_equal(pmaps[k].cpu().numpy(), want[k & 1][1]["pmap"]), k
    got = grid.read(0, want=("pmap", "pass", "hit"))          # a main-stream map call: joins the map stream
    w = want[1]
    assert np.array_equal(got["pass"], w[1]["pass"]) and np.array_equal(got["hit"], w[1]["hit"]) and grid.visits() == w[2]
    assert np.array_equal(bufs[0].cpu().numpy(), w[0][0])
    assert np.array_equal(tbufs[0].cpu().numpy().reshape(w[0][1].shape), w[0][1])
    # map calls on the main stream between pipelined replays keep their order
    m_ox = np.array([[1.0, 2.0, -3.0]]); m_oy = np.array([[0.5, -1.0, 2.0]])
    grid.reset()
    grid.update_host(m_ox, m_oy, np.array([0.0]), np.array([0.0]))
    c1 = grid.read(0, want=("pass", "hit"))
    grid.reset()
    c0 = grid.read(0, want=("pass", "hit"))
    assert c1["pass"].sum() > 0 and c1["hit"].sum() == 3 and c0["pass"].sum() == 0 and c0["hit"].sum() == 0
    ctx.set_option("pipeline", 0)


@pytest.mark.parametrize("group", [1, 2, 3, 8, 11, 64])
def test_grid_window_modes_are_bit_identical(slam, syn, group):
    """The LDS-window ray caster (any group size, incl. windows smaller than the rays'
    bounding box) and the direct-atomic one produce the same counters as the oracle."""
    rep = syn.make_replay(40, 360, seed=2, stride=5, room_scale=2.0)      # 20 x 16 m room: rays leave a 192-cell window
    ctx = slam.Context(0)
    og = co.Grid(2000, 2000, 50.0, 20.0, 20.0)
    oposes, _, _, ov = co.replay(rep.ranges, AMIN, AMAX, og, threads=8, mt_grid=True)
    for mode, split in ((0, -1), (1, -1), (2, -1), (3, 0), (3, 1), (4, -1)):   # direct atomics, automatic (= wedges on a map this
        ctx.set_option("grid_mode", mode)                                       # large), tiles, window with one / two workgroups
        ctx.set_option("grid_group", group)                                     # per group of scans, wedges
        ctx.set_option("grid_split", split)
        grid = slam.DeviceGrid.metric(1, 2000, 2000, 0.02, context=ctx)
        poses, _, _ = slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid, context=ctx)
        r = grid.read(0, want=("pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt), (mode, group)
        assert grid.visits() == ov
        grid.close()
    # explicit-endpoint form (Mapping.update's arguments), many scans into one map
    pts = [co.laser_to_points(r, AMIN, AMAX) for r in rep.ranges[1:]]
    wp = [co.world_points(p, x, y) for p, (x, y) in zip(oposes, pts)]
    ox, oy = np.stack([w[0] for w in wp]), np.stack([w[1] for w in wp])
    ctx.set_option("grid_mode", 1)
    ctx.set_option("grid_split", -1)
    grid = slam.DeviceGrid.metric(1, 2000, 2000, 0.02, context=ctx)
    grid.update_host(ox, oy, oposes[:, 0], oposes[:, 1])
    r = grid.read(0, want=("pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt)
    grid.close()
    ctx.close()


@pytest.mark.parametrize("split", [0, 1])
def test_window_kernel_one_or_two_workgroups_per_group(slam, syn, split):
    """The window ray cast with one workgroup per group of scans, or two - one per direction half (a ray never crosses
    the column of its origin) - on the benchmark's map: 120 scans in groups of 1, 7 and 12 (ragged last group): same
    counters, same visits."""
    rep = syn.make_replay(121, 360, seed=6, stride=5)
    og = co.Grid(400, 400, 20.0, 10.0, 10.0)
    _, _, _, ov = co.replay(rep.ranges, AMIN, AMAX, og, threads=8, mt_grid=True)
    for group in (1, 7, 12):
        ctx = slam.Context(0)
        ctx.set_option("grid_mode", 3)
        ctx.set_option("grid_group", group)
        ctx.set_option("grid_split", split)
        grid = slam.DeviceGrid.metric(1, 400, 400, 0.05, context=ctx)
        slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid, context=ctx)
        r = grid.read(0, want=("pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and grid.visits() == ov, group
        grid.close()
        ctx.close()


@pytest.mark.parametrize("mode", [2, 4])
def test_tiled_ray_cast_cases(slam, syn, mode):
    """The two paths for maps much larger than a window - recorded walks + tiles (grid_mode 2), direction wedges swept
    in bands (grid_mode 4) - beyond the common case: several streams into one map, a small map (path forced), rays
    longer than the 2048 recorded steps / than a band, rays leaving the map, and separate ray origins."""
    ctx = slam.Context(0)
    ctx.set_option("grid_mode", mode)
    # (a) three streams of 12 scans into one 400x400 map
    reps = [syn.make_replay(13, 200, seed=s, stride=5) for s in (7, 8, 9)]
    ranges = np.stack([r.ranges for r in reps])
    og = co.Grid(400, 400, 20.0, 10.0, 10.0)
    ov = 0
    for r in ranges:
        _, _, _, v = co.replay(r, AMIN, AMAX, og, threads=8, mt_grid=True)
        ov += v
    for group in (0, 5):                                     # 5 does not divide 12: the group size is lowered
        ctx.set_option("grid_group", group)
        grid = slam.DeviceGrid.metric(1, 400, 400, 0.05, context=ctx)
        slam.replay_host(ranges, AMIN, AMAX, grid=grid, grid_of_traj=None, context=ctx)
        r = grid.read(0, want=("pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and grid.visits() == ov
        grid.close()
    ctx.set_option("grid_group", 0)
    # (b) 0.004 m cells: most rays are longer than 2048 steps; the map (3000^2) holds only part of the room
    rep = syn.make_replay(9, 120, seed=11, stride=5)
    og = co.Grid(3000, 3000, 250.0, 6.0, 6.0)
    oposes, _, _, ov = co.replay(rep.ranges, AMIN, AMAX, og, threads=8, mt_grid=True)
    grid = slam.DeviceGrid(1, 3000, 3000, 250.0, 6.0, 6.0, context=ctx)
    slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid, context=ctx)
    r = grid.read(0, want=("pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and grid.visits() == ov
    assert int(r["pass"].max()) > 0
    grid.close()
    # (c) separate ray origins (slam_grid_update_scans) through the tiled path
    rep = syn.make_replay(12, 200, seed=12, stride=5)
    poses = np.ascontiguousarray(rep.poses_true[:12])
    centres = poses[:, :2] + np.random.default_rng(4).normal(0, 0.05, size=(12, 2))
    m = slam.Mapping(200, 200, 0.1, context=ctx)
    m.update_scans(rep.ranges[:12], AMIN, AMAX, poses, centres)
    o = on.Mapping(200, 200, 0.1)
    for k in range(12):
        obs = on.world_points(poses[k], on.laser_to_numpy(rep.ranges[k], AMIN, AMAX, clip_inf=True))
        o.update(obs[0], obs[1], centres[k][0], centres[k][1])
    p, h = m.counters()
    assert np.array_equal(p, o.pass_cnt) and np.array_equal(h, o.hit_cnt)
    # (d) explicit world-frame endpoints (Mapping.update's arguments), 7 scans in one call
    rng = np.random.default_rng(9)
    cx, cy = rng.uniform(-3, 3, 7), rng.uniform(-3, 3, 7)
    ang, d = rng.uniform(-np.pi, np.pi, (7, 150)), rng.uniform(0, 14, (7, 150))
    ox, oy = cx[:, None] + np.cos(ang) * d, cy[:, None] + np.sin(ang) * d
    ox[2, :9] = np.inf
    g = slam.DeviceGrid(1, 200, 240, 10.0, 10.0, 12.0, context=ctx)
    g.update_host(ox, oy, cx, cy)
    og = co.Grid(200, 240, 10.0, 10.0, 12.0)
    for b in range(7):
        og.update(ox[b], oy[b], cx[b], cy[b])
    r = g.read(0, want=("pmap", "pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap)
    assert g.visits() == og.visits
    g.close()
    # (e) scans of ONE group cast from origins 38 000 cells apart on a long narrow map: rays of one direction class that no
    #     window can hold together (the wedges' bands fall back to direct atomics; the tiles do not care)
    cx, cy = np.array([0.3, -0.4, 0.1]), np.array([-950.0, 955.0, 10.0])
    ang, d = rng.uniform(-np.pi, np.pi, (3, 200)), rng.uniform(0.2, 9.0, (3, 200))
    ox, oy = cx[:, None] + np.cos(ang) * d, cy[:, None] + np.sin(ang) * d
    g = slam.DeviceGrid(1, 400, 40000, 20.0, 10.0, 1000.0, context=ctx)
    g.update_host(ox, oy, cx, cy)
    og = co.Grid(400, 40000, 20.0, 10.0, 1000.0)
    for b in range(3):
        og.update(ox[b], oy[b], cx[b], cy[b])
    r = g.read(0, want=("pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and g.visits() == og.visits
    assert int(r["pass"].sum()) > 10000
    g.close()
    ctx.close()


def test_particle_hypotheses_pipeline_vs_oracle(slam, syn):
    """BASELINE configs[2] operator: P perturbed priors of one scan pair, one map per
    hypothesis.  Oracle = loop over hypotheses (ICP on the perturbed source, one
    dead-reckoning step with M = T.prior, ray cast of the original scan)."""
    rep = syn.make_replay(2, 360, seed=2, stride=5)
    P = 24
    mats = slam.prior_matrices(syn.particle_priors(P, seed=2))
    pose_prev = np.random.default_rng(4).normal(0, 0.5, size=(P, 3))
    grid = slam.DeviceGrid.metric(P, 400, 400, 0.05)
    poses, T, it = slam.particles_host(rep.ranges[0], rep.ranges[1], AMIN, AMAX, mats, pose_prev, grid=grid)
    tar = np.array(co.laser_to_points(rep.ranges[0], AMIN, AMAX))
    src = np.array(co.laser_to_points(rep.ranges[1], AMIN, AMAX))
    for p in range(P):
        m = mats[p]
        sp = np.stack([m[0, 0] * src[0] + m[0, 1] * src[1] + m[0, 2], m[1, 0] * src[0] + m[1, 1] * src[1] + m[1, 2]])
        oT, oit, _ = co.icp_process(tar, sp, 30, 0.001)
        M = np.eye(3)
        M[0, 0] = oT[0, 0] * m[0, 0] + oT[0, 1] * m[1, 0]; M[1, 0] = oT[1, 0] * m[0, 0] + oT[1, 1] * m[1, 0]
        M[0, 2] = oT[0, 0] * m[0, 2] + oT[0, 1] * m[1, 2] + oT[0, 2]; M[1, 2] = oT[1, 0] * m[0, 2] + oT[1, 1] * m[1, 2] + oT[1, 2]
        op = co.compose_pose(pose_prev[p], M)
        assert it[p] == oit and np.max(np.abs(T[p] - oT)) < FTOL and np.max(np.abs(poses[p] - op)) < FTOL, p
        og = co.Grid(400, 400, 20.0, 10.0, 10.0)
        ox, oy = co.world_points(op, src[0], src[1])
        og.update(ox, oy, op[0], op[1])
        r = grid.read(p, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap), p
    assert len(set(it.tolist())) > 1


@pytest.mark.parametrize("P", [5, 64])
def test_particle_hypotheses_without_priors(slam, syn, P):
    """slam_particles with prior == NULL and a small batch: the ray cast reads cos / sin of the new headings from the
    pose step's scratch, so the pose step must be the kernel that writes them whatever the batch size (until round 4 a
    batch of <= 64 hypotheses without priors took k_pose_compose, which does not, and the maps were cast with stale
    scratch bytes for a heading)."""
    rep = syn.make_replay(2, 360, seed=5, stride=5)
    pose_prev = np.random.default_rng(6).normal(0, 0.5, size=(P, 3))
    grid = slam.DeviceGrid.metric(P, 400, 400, 0.05)
    # (a second batch right behind one WITH priors, whose headings differ: stale scratch would show)
    slam.particles_host(rep.ranges[0], rep.ranges[1], AMIN, AMAX, slam.prior_matrices(syn.particle_priors(P, seed=3)), pose_prev + 1.0, grid=None)
    poses, T, it = slam.particles_host(rep.ranges[0], rep.ranges[1], AMIN, AMAX, None, pose_prev, grid=grid)
    tar = np.array(co.laser_to_points(rep.ranges[0], AMIN, AMAX))
    src = np.array(co.laser_to_points(rep.ranges[1], AMIN, AMAX))
    oT, oit, _ = co.icp_process(tar, src, 30, 0.001)
    for p in range(P):
        op = co.compose_pose(pose_prev[p], oT)
        assert it[p] == oit and np.max(np.abs(T[p] - oT)) < FTOL and np.max(np.abs(poses[p] - op)) < FTOL, p
        og = co.Grid(400, 400, 20.0, 10.0, 10.0)
        ox, oy = co.world_points(op, src[0], src[1])
        og.update(ox, oy, op[0], op[1])
        r = grid.read(p, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap), p
    grid.close()


def test_wedge_sort_lds_need_grows_within_one_process(slam):
    """k_wedge_sort's dynamic LDS depends on beams x scans per group: a cast with few beams (attribute set to a few KB)
    followed by one that needs more than the 64 KB default must raise the attribute again (allow_dynamic_lds remembers
    the largest size per kernel, not just that it was set)."""
    ctx = slam.Context(0)
    ctx.set_option("grid_mode", 4)
    rng = np.random.default_rng(21)
    for B, n in ((16, 96), (16, 2040), (3, 8192)):               # 16 x 2040 rays: 2 x 32 640 B of keys + bins > 64 KB
        cx, cy = rng.uniform(-2, 2, B), rng.uniform(-2, 2, B)
        ang, d = rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(0.1, 9.0, (B, n))
        ox, oy = cx[:, None] + np.cos(ang) * d, cy[:, None] + np.sin(ang) * d
        g = slam.DeviceGrid(1, 300, 280, 12.5, 12.0, 11.2, context=ctx)
        g.update_host(ox, oy, cx, cy)
        og = co.Grid(300, 280, 12.5, 12.0, 11.2)
        for b in range(B):
            og.update(ox[b], oy[b], cx[b], cy[b])
        r = g.read(0, want=("pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and g.visits() == og.visits, (B, n)
        g.close()
    ctx.close()


@pytest.mark.parametrize("mode", [2, 4])
def test_forced_tiled_modes_keep_the_beam_bound(slam, mode):
    """grid_mode 2 / 4 force the tiled / wedge path only up to 8192 beams per scan (ray numbers inside a group are
    16-bit); a scan with more beams takes the window / direct path instead of being cast with truncated ray numbers."""
    ctx = slam.Context(0)
    ctx.set_option("grid_mode", mode)
    rng = np.random.default_rng(22)
    for n in (8193, 70001):
        ang, d = rng.uniform(-np.pi, np.pi, n), rng.uniform(0.1, 9.0, n)
        ox, oy = 0.3 + np.cos(ang) * d, -0.2 + np.sin(ang) * d
        g = slam.DeviceGrid(1, 200, 200, 10.0, 10.0, 10.0, context=ctx)
        g.update_host(ox, oy, 0.3, -0.2)
        og = co.Grid(200, 200, 10.0, 10.0, 10.0)
        og.update(ox, oy, 0.3, -0.2)
        r = g.read(0, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap), n
        assert g.visits() == og.visits
        g.close()
    ctx.close()


@pytest.mark.parametrize("n", [120, 360])
@pytest.mark.parametrize("offset", [(1000.0, -700.0), (-3.0e4, 2.5e4)])
def test_icp_clouds_far_from_the_origin_one_pass_products(slam, syn, n, offset):
    """From its second iteration on k_icp forms the centred cross-covariance in ONE pass about the previous matches'
    centroid p, W = sum (b - p)(a - p)^T - S_b S_a^T / N, where the reference centres first (icp.py:154-160).  The two
    differ by roundings of the order of W's last place as long as p is close to the centroids - which the
    coordinates' magnitude must not change: clouds 1e3 and 4e4 m from the origin (coordinates 200 to 8000 times the
    clouds' spread), iteration counts equal and transforms to 1e-9 (relative to the offset for the translation)."""
    reps = [syn.make_replay(9, n, seed=40 + s, stride=5) for s in range(3)]
    tars, srcs = [], []
    for rep in reps:
        pts = np.stack([np.array(co.laser_to_points(r, AMIN, AMAX)) for r in rep.ranges])
        pts[:, 0, :] += offset[0]
        pts[:, 1, :] += offset[1]
        tars.append(pts[:-1]); srcs.append(pts[1:])
    tars, srcs = np.concatenate(tars), np.concatenate(srcs)
    T, it, err = slam.icp_batch_host(tars, srcs, 30, 0.001)
    oT, oit, oerr = co.icp_batch(tars, srcs, 30, 0.001)
    assert np.array_equal(it, oit), (it, oit)
    assert int(it.max()) >= 3                                    # the one-pass iterations really ran
    scale = max(abs(offset[0]), abs(offset[1]))
    assert np.max(np.abs(T[:, :2, :2] - oT[:, :2, :2])) < FTOL
    assert np.max(np.abs(T[:, :2, 2] - oT[:, :2, 2])) < FTOL * scale
    assert np.max(np.abs(err - oerr)) < FTOL


@pytest.mark.parametrize("max_iter", [1, 2, 30])
def test_icp_final_transform_offset_million_times_spread(slam, syn, max_iter):
    """the final T is rebuilt in ONE reduction from the first iteration's source centroid and the last iteration's
    match centroid (k_icp; the reference calls getTransform(A_original, src_final) with two-pass centroids, icp.py:81,
    154-160), dropping a term of the order rounding x rounding.  Where that matters most: clouds a MILLION times their
    spread from the origin (offset / spread >= 1e6: 1e6 ... 4e6 m for a 1 ... 4 m cloud), with a prior applied to the source
    and with max_iter 1 and 2, where the final transform follows the very first update.  Iteration counts equal; rotation to
    1e-9; translation to 1e-9 of the offset (a coordinate's own last place there is 2e-10 m; the oracle loses the same digits)."""
    reps = [syn.make_replay(7, 120, seed=60 + s, stride=5) for s in range(3)]
    off = np.array([1.0e6, -4.0e6])
    tars, srcs = [], []
    rng = np.random.default_rng(61)
    for rep in reps:
        pts = np.stack([np.array(co.laser_to_points(r, AMIN, AMAX)) for r in rep.ranges]) * 0.25       # spread ~1-2 m
        tars.append(pts[:-1] + off[None, :, None])
        th, tr = rng.normal(0, 0.02, len(pts) - 1), rng.normal(0, 0.03, (len(pts) - 1, 2))
        src = pts[1:]
        c, sn = np.cos(th)[:, None], np.sin(th)[:, None]
        srcs.append(np.stack([c * src[:, 0] - sn * src[:, 1] + tr[:, 0:1], sn * src[:, 0] + c * src[:, 1] + tr[:, 1:2]], axis=1) + off[None, :, None])
    tars, srcs = np.concatenate(tars), np.concatenate(srcs)
    T, it, err = slam.icp_batch_host(tars, srcs, max_iter, 0.001)
    oT, oit, oerr = co.icp_batch(tars, srcs, max_iter, 0.001)
    assert np.array_equal(it, oit), (it, oit)
    assert np.max(np.abs(T[:, :2, :2] - oT[:, :2, :2])) < FTOL
    assert np.max(np.abs(T[:, :2, 2] - oT[:, :2, 2])) < FTOL * np.abs(off).max()
    assert np.max(np.abs(err - oerr)) < 1e-7                         # (mean distance of points whose coordinates carry 2e-10 m)
    # the transform maps the source onto the target about as well as the oracle's does (a wrong centroid would show as metres)
    for b in (0, len(T) - 1):
        r_dev = T[b, :2, :2] @ srcs[b] + T[b, :2, 2:3]
        r_ref = oT[b, :2, :2] @ srcs[b] + oT[b, :2, 2:3]
        assert np.max(np.abs(r_dev - r_ref)) < 1e-4


def test_particle_batch_in_chunks_is_identical(slam, syn):
    """Context option "particle_chunks": the batch is cut into chunks whose ray casts run on a second stream behind the
    next chunk's scan matching.  Hypotheses are independent, so poses, transforms, iteration counts and every map must
    equal the one-piece batch bit for bit, a second batch into settled maps included.  (The host-pointer path synchronises
    between batches; the device path without any host synchronise is the next test.)"""
    rep = syn.make_replay(2, 360, seed=2, stride=5)
    P = 50
    mats = slam.prior_matrices(syn.particle_priors(P, seed=8))
    pose_prev = np.random.default_rng(9).normal(0, 0.5, size=(P, 3))
    out = {}
    for chunks in (1, 3, 7):
        ctx = slam.Context(0)
        ctx.set_option("particle_chunks", chunks)
        grid = slam.DeviceGrid.metric(P, 400, 400, 0.05, context=ctx)
        grid.live_pmap()
        for _ in range(2):                                       # the second batch meets settled maps
            poses, T, it = slam.particles_host(rep.ranges[0], rep.ranges[1], AMIN, AMAX, mats, pose_prev, grid=grid, context=ctx)
        maps = [grid.read(p, want=("pmap", "pass", "hit")) for p in range(P)]
        out[chunks] = (poses, T, it, maps, grid.visits())
        grid.close()
        ctx.close()
    for chunks in (3, 7):
        a, b = out[1], out[chunks]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[4] == b[4]
        for p in range(P):
            for k in ("pmap", "pass", "hit"):
                assert np.array_equal(a[3][p][k], b[3][p][k]), (chunks, p, k)


@pytest.mark.parametrize("chunks,pipeline", [(3, 0), (4, 1), (2, 1)])
def test_particles_dev_back_to_back_without_host_sync(slam, syn, chunks, pipeline):
    """with "particle_chunks" > 1 the ray casts of a batch run on a second stream and read the CALLER's buffers
    (ranges2, trig tables, poses_out) and the context's heading scratch.  slam_particles_dev is a *_dev entry point: work
    enqueued behind it must be ordered behind all of it.  Here, with NO host synchronise anywhere between the calls:
    batch 1 on scan pair (0, 1) -> ranges2 overwritten on the context's stream with pair (1, 2) -> a slam_replay_dev with
    T_out == NULL (it carves the same scratch arena the headings live in) -> batch 2 on the new pair from batch 1's poses
    into the same maps.  Everything - both batches' poses, transforms, iteration counts, every map - must equal the
    one-piece, one-stream run bit for bit (hypotheses are independent: ICP.process / Mapping.update per hypothesis)."""
    import torch
    A = slam._abi
    L = A.lib()
    rep = syn.make_replay(3, 360, seed=2, stride=5)
    small = syn.make_replay(40, 360, seed=12, stride=5)
    P = 60          # (at most 64 pairs a launch, whole or in chunks: one launch shape, so the sums are formed in one order and the floats agree bit for bit)
    mats = slam.prior_matrices(syn.particle_priors(P, seed=8)).reshape(P, 6)
    pose_prev = np.random.default_rng(9).normal(0, 0.3, size=(P, 3))
    ct, sn = A.trig_tables(AMIN, AMAX, 360)

    def run(chunks, pipeline):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            ctx = A.Context(0, st.cuda_stream)
            ctx.set_option("pipeline", pipeline)
            ctx.set_option("particle_chunks", chunks)
            grid = slam.DeviceGrid.metric(P, 400, 400, 0.05, context=ctx)
            grid.live_pmap()
            r2, r2_next = d(rep.ranges[0:2].astype(np.float32)), d(rep.ranges[1:3].astype(np.float32))
            cos_t, sin_t, prior, p0 = d(ct), d(sn), d(mats), d(pose_prev)
            sm_r, sm_p0 = d(small.ranges.astype(np.float32)[None]), d(np.zeros((1, 3)))
            sm_poses = torch.empty((1, 39, 3), dtype=torch.float64, device="cuda")
            o = [dict(poses=torch.empty((P, 3), dtype=torch.float64, device="cuda"), T=torch.empty((P, 9), dtype=torch.float64, device="cuda"),
                      it=torch.empty(P, dtype=torch.int32, device="cuda")) for _ in range(2)]
            st.synchronize()

            def batch(k, prev):
                A.check(L.slam_particles_dev(ctx.handle, r2.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), 360, A.F64, prior.data_ptr(),
                                             prev.data_ptr(), P, 30, 1e-3, grid._h, None, o[k]["poses"].data_ptr(), o[k]["T"].data_ptr(),
                                             o[k]["it"].data_ptr()))
            batch(0, p0)
            r2.copy_(r2_next)                                        # on the context's stream (= torch's current one here)
            A.check(L.slam_replay_dev(ctx.handle, sm_r.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), 1, 40, 360, A.F64, 30, 1e-3,
                                      sm_p0.data_ptr(), None, None, None, sm_poses.data_ptr(), None, None))
            batch(1, o[0]["poses"])
            ctx.synchronize()
            ctx.check_status()
            res = [{k: v.cpu().numpy() for k, v in ob.items()} for ob in o]
            res.append(sm_poses.cpu().numpy())
            maps = [grid.read(p, want=("pmap", "pass", "hit")) for p in range(0, P, 5)]
            vis = grid.visits()
            grid.close()
            ctx.close()
        return res, maps, vis

    want, wmaps, wvis = run(1, 0)
    got, gmaps, gvis = run(chunks, pipeline)
    for k in range(2):
        for key in ("poses", "T", "it"):
            assert np.array_equal(want[k][key], got[k][key]), (k, key)
    assert np.array_equal(want[2], got[2]) and wvis == gvis
    assert int(want[0]["it"].max()) >= 3 and not np.array_equal(want[0]["poses"], want[1]["poses"])
    for a, b in zip(wmaps, gmaps):
        for key in ("pmap", "pass", "hit"):
            assert np.array_equal(a[key], b[key]), key
