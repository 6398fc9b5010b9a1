"""Pins the CPU oracle (oracle/oracle_np.py and oracle/slam_oracle.c) to golden vectors
produced by the reference's own code (oracle/gen_golden.py).  CPU only.

Tolerances: integer / index / cell results bit-exact; float64 results 1e-9 absolute
(the restatements differ from NumPy/LAPACK only in summation order and in using a
closed-form 2x2 SVD)."""
import hashlib
import zlib

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import oracle_np as on

FTOL = 1e-9


def _path_bytes(p):
    return np.asarray(p, dtype="<i4").reshape(-1).tobytes()


# ---------------------------------------------------------------- G1 bresenham
@pytest.mark.parametrize("impl", ["np", "c"])
def test_bresenham_fan(g1, impl):
    f = on.bresenham_path if impl == "np" else co.bresenham
    ends, cells, offs = g1["fan_ends"], g1["fan_cells"], g1["fan_offsets"]
    for k in range(ends.shape[0]):
        want = cells[offs[k]:offs[k + 1]].astype(np.int32)
        got = np.asarray(f([0, 0], [int(ends[k, 0]), int(ends[k, 1])]), dtype=np.int32).reshape(-1, 2)
        assert got.shape == want.shape and np.array_equal(got, want), ends[k]


@pytest.mark.parametrize("impl", ["np", "c"])
def test_bresenham_random_long(g1, impl):
    f = on.bresenham_path if impl == "np" else co.bresenham
    sha = hashlib.sha256()
    n = 2000 if impl == "c" else 300
    for k in range(n):
        p = np.asarray(f(g1["rand_starts"][k], g1["rand_ends"][k]), dtype=np.int32).reshape(-1, 2)
        assert len(p) == g1["rand_len"][k]
        b = _path_bytes(p) if len(p) else b""
        assert zlib.crc32(b) == g1["rand_crc"][k], k
        if len(p):
            assert np.array_equal(p[0], g1["rand_first_last"][k, 0]) and np.array_equal(p[-1], g1["rand_first_last"][k, 1])
        sha.update(b)
    if n == 2000:
        assert np.array_equal(np.frombuffer(sha.digest(), dtype=np.uint8), g1["rand_sha256"])


def test_bresenham_not_integer_bresenham(g1):
    """the float error term makes a share of lines differ from the
    textbook integer algorithm; the oracle must follow the float one."""
    def integer_line(dx, dy):
        y, err, out = 0, 0, []
        for x in range(dx + 1):
            out.append((x, y))
            err += 2 * dy
            if err >= dx:
                y += 1
                err -= 2 * dx
        return out
    differ = sum(on.bresenham_path([0, 0], [dx, dy]) != integer_line(dx, dy)
                 for dx in range(1, 60) for dy in range(0, dx + 1))
    assert differ > 50


# ---------------------------------------------------------------- G2 mapping
def _run_np(xw, yw, oxs, oys, cs):
    m = on.Mapping(xw, yw, 0.1)
    steps = []
    for ox, oy, c in zip(oxs, oys, cs):
        steps.append(m.update(ox, oy, c[0], c[1]).astype(np.int8).copy())
    return m, steps


def _run_c(xw, yw, oxs, oys, cs):
    m = co.Grid(xw, yw)
    steps = []
    for ox, oy, c in zip(oxs, oys, cs):
        steps.append(m.update(ox, oy, c[0], c[1]).copy())
    return m, steps


@pytest.mark.parametrize("impl", ["np", "c"])
@pytest.mark.parametrize("n", [120, 200, 360])
def test_mapping_demo(g2, impl, n):
    run = _run_np if impl == "np" else _run_c
    m, steps = run(200, 200, g2["demo%d_ox" % n], g2["demo%d_oy" % n], g2["demo%d_c" % n])
    assert np.array_equal(np.array(steps), g2["demo%d_pmap_steps" % n])
    assert np.array_equal(np.asarray(m.datamap), g2["demo%d_datamap" % n])   # same order of float adds: exact


@pytest.mark.parametrize("impl", ["np", "c"])
def test_mapping_static_centre_saturation(g2, impl):
    run = _run_np if impl == "np" else _run_c
    c = g2["static_c"]
    m, steps = run(200, 200, g2["static_ox"], g2["static_oy"], [c] * 5)
    assert np.array_equal(np.array(steps), g2["static_pmap_steps"])
    assert np.array_equal(np.asarray(m.datamap), g2["static_datamap"])
    cx, cy = int(10 * (c[0] + 10)), int(10 * (c[1] + 10))
    # 360 rays/scan * 0.01 = 3.6 per scan: free after 2 scans, "occupied" after 3 (SURVEY a-10)
    assert [int(s[cx, cy]) for s in steps] == [0, 0, 100, 100, 100]


@pytest.mark.parametrize("impl", ["np", "c"])
@pytest.mark.parametrize("case,xw,yw", [("edge", 200, 200), ("outside", 200, 200), ("rect", 120, 260)])
def test_mapping_cases(g2, impl, case, xw, yw):
    run = _run_np if impl == "np" else _run_c
    m, steps = run(xw, yw, [g2[case + "_ox"]], [g2[case + "_oy"]], [g2[case + "_c"]])
    assert np.array_equal(steps[-1], g2[case + "_pmap"])
    assert np.array_equal(np.asarray(m.datamap), g2[case + "_datamap"])


@pytest.mark.parametrize("n", [120, 360])
def test_integer_counter_rule_equals_float_threshold(g2, n):
    """The rule the HIP finalize kernel uses: untouched 50; hit>=1 or pass>=1001 -> 100."""
    assert on.pass_count_threshold() == 1001 == co.pass_count_threshold()
    m, steps = _run_np(200, 200, g2["demo%d_ox" % n], g2["demo%d_oy" % n], g2["demo%d_c" % n])
    assert np.array_equal(m.pmap_from_counts(), g2["demo%d_pmap_steps" % n][-1])
    mc, _ = _run_c(200, 200, g2["demo%d_ox" % n], g2["demo%d_oy" % n], g2["demo%d_c" % n])
    assert np.array_equal(mc.pass_cnt, m.pass_cnt) and np.array_equal(mc.hit_cnt, m.hit_cnt)
    approx = 0.01 * m.pass_cnt + 20.0 * m.hit_cnt
    assert np.max(np.abs(approx - g2["demo%d_datamap" % n])) < 1e-9


def test_pass_threshold_saturation_long_run():
    """1000 passes leave the cell free, the 1001st flips it (sequential float64 sum)."""
    m = on.Mapping(200, 200, 0.1)
    for k in range(1001):
        m.update(np.array([0.55]), np.array([0.0]), 0.0, 0.0)   # 5-cell ray, 4 pass cells + 1 hit
        if k == 999:
            assert m.pmap[101, 100] == 0 and m.pass_cnt[101, 100] == 1000
    assert m.pmap[101, 100] == 100 and m.pass_cnt[101, 100] == 1001
    assert np.array_equal(m.pmap_from_counts(), m.pmap.astype(np.int8))


# ---------------------------------------------------------------- G3 icp
def _lt(ranges):
    return on.laser_to_numpy(ranges, -3.14159, 3.14159, clip_inf=True)


def test_laser_to_numpy(g3):
    for k in range(int(g3["nn_count"])):
        r = g3["nn%d_ranges" % k]
        assert np.array_equal(_lt(r[0]), g3["nn%d_tar" % k]) and np.array_equal(_lt(r[1]), g3["nn%d_src" % k])
        x, y = co.laser_to_points(r[1], -3.14159, 3.14159)
        assert np.array_equal(x, g3["nn%d_src" % k][0]) and np.array_equal(y, g3["nn%d_src" % k][1])


@pytest.mark.parametrize("impl", ["np_loop", "np", "c"])
def test_find_nearest(g3, impl):
    f = {"np_loop": on.find_nearest_loop, "np": on.find_nearest, "c": co.find_nearest}[impl]
    for k in range(int(g3["nn_count"])):
        tar, src = g3["nn%d_tar" % k], g3["nn%d_src" % k]
        if impl == "np_loop" and tar.shape[1] > 120:
            continue
        d, i = f(src[:2].T.copy(), tar[:2].T.copy())
        assert np.array_equal(np.asarray(i, dtype=np.int32), g3["nn%d_idx" % k])
        assert np.max(np.abs(d - g3["nn%d_dist" % k])) < 1e-15


def test_find_nearest_ties_and_nan():
    tar = np.array([[1.0, 0.0], [-1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    src = np.array([[0.0, 0.0], [np.nan, 0.0], [1.0, 0.0]])
    for f in (on.find_nearest_loop, on.find_nearest, co.find_nearest):
        d, i = f(src, tar)
        assert list(i) == [0, 0, 0] and d[0] == 1.0 and d[1] == 0.0 and d[2] == 0.0


@pytest.mark.parametrize("impl", ["np", "np_closed", "c"])
def test_get_transform(g3, impl):
    f = {"np": on.get_transform, "np_closed": on.get_transform_closed_form, "c": co.get_transform}[impl]
    assert int(g3["gt_reflect"].sum()) == 50
    for k in range(100):
        a, b = g3["gt_src"][k], g3["gt_tar"][k]
        n = int(np.sum(~np.isnan(a[:, 0])))
        T = f(a[:n].copy(), b[:n].copy())
        assert np.max(np.abs(T - g3["gt_T"][k])) < FTOL, (k, bool(g3["gt_reflect"][k]))


@pytest.mark.parametrize("impl", ["np", "c"])
def test_icp_process(g3, impl):
    for k in range(g3["pr_T"].shape[0]):
        n = int(g3["pr_n"][k])
        mi, tol = int(g3["pr_cfg"][k, 0]), float(g3["pr_cfg"][k, 1])
        tar, src = _lt(g3["pr_ranges"][k, 0, :n]), _lt(g3["pr_ranges"][k, 1, :n])
        if impl == "np":
            T, it, me = on.icp_process(tar, src, mi, tol, return_info=True)
        else:
            T, it, me = co.icp_process(tar, src, mi, tol)
        assert it == g3["pr_iters"][k], k
        assert np.max(np.abs(T - g3["pr_T"][k])) < FTOL, k
        assert abs(me - g3["pr_mean_err"][k]) < FTOL


def test_icp_iteration_counts_cover_both_exits(g3):
    it, cfg = g3["pr_iters"], g3["pr_cfg"]
    assert np.all(it[cfg[:, 1] == 0.0] == 10)                 # tol 0: never converges, runs max_iter
    assert np.all(it[cfg[:, 1] > 0] < 30) and it.min() >= 2   # default: early exit


@pytest.mark.parametrize("impl", ["np", "c"])
def test_icp_ragged_sizes(g3, impl):
    tar, src = _lt(g3["rag_r0"]), _lt(g3["rag_r1"])
    assert tar.shape[1] == 120 and src.shape[1] == 60 and np.hypot(tar[0, 5], tar[1, 5]) == 30.0
    if impl == "np":
        T, it, _ = on.icp_process(tar, src, 30, 0.001, return_info=True)
    else:
        T, it, _ = co.icp_process(tar, src, 30, 0.001)
    assert it == int(g3["rag_iters"]) and np.max(np.abs(T - g3["rag_T"])) < FTOL


# ---------------------------------------------------------------- G4 pipeline
@pytest.mark.parametrize("impl", ["np", "c"])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_pipeline(g4, impl, tag):
    ranges = g4[tag + "_ranges"]
    if impl == "np":
        m = on.Mapping(200, 200, 0.1)
        poses, T, _ = on.replay(ranges, -3.14159, 3.14159, m)
        pmap, datamap, data = m.pmap.astype(np.int8), m.datamap, on.occupancy_grid_data(m.pmap)
    else:
        m = co.Grid(200, 200)
        poses, T, _, _ = co.replay(ranges, -3.14159, 3.14159, m)
        pmap, datamap, data = m.pmap, m.datamap, m.occupancy_grid_data()
    assert np.max(np.abs(poses - g4[tag + "_poses"])) < FTOL
    assert np.max(np.abs(T - g4[tag + "_T"])) < FTOL
    assert np.array_equal(pmap, g4[tag + "_pmap"])
    assert np.array_equal(data, g4[tag + "_grid_data"])
    assert np.max(np.abs(datamap - g4[tag + "_datamap"])) < 1e-9


def test_c_replay_threads_are_deterministic(g4):
    r = g4["a_ranges"]
    p1, T1, i1, v1 = co.replay(r, -3.14159, 3.14159, co.Grid(200, 200), threads=1)
    p4, T4, i4, v4 = co.replay(r, -3.14159, 3.14159, co.Grid(200, 200), threads=4)
    assert np.array_equal(p1, p4) and np.array_equal(T1, T4) and np.array_equal(i1, i4) and v1 == v4


def test_c_replay_mt_equals_serial(syn):
    """The all-cores baseline (orc_replay_mt) produces the serial oracle's poses, counters and pmap."""
    rep = syn.make_replay(40, 120, seed=4, stride=5)
    g1_, g2_ = co.Grid(400, 400, 20.0, 10.0, 10.0), co.Grid(400, 400, 20.0, 10.0, 10.0)
    p1, T1, i1, v1 = co.replay(rep.ranges, -3.14159, 3.14159, g1_, threads=1)
    p2, T2, i2, v2 = co.replay(rep.ranges, -3.14159, 3.14159, g2_, threads=8, mt_grid=True)
    assert np.array_equal(p1, p2) and np.array_equal(i1, i2) and v1 == v2
    assert np.array_equal(g1_.pass_cnt, g2_.pass_cnt) and np.array_equal(g1_.hit_cnt, g2_.hit_cnt)
    assert np.array_equal(g1_.pmap, g2_.pmap)


# ---------------------------------------------------------------- G5 scan-to-map observation
@pytest.fixture(scope="module")
def g5():
    from conftest import load_golden
    return load_golden("g5_map_observation.npz")


@pytest.mark.parametrize("impl", ["np", "c"])
def test_map_obstacles(g5, impl):
    f = on.map_obstacles if impl == "np" else co.map_obstacles
    obs = f(g5["map_data"], 200, 200, 0.1, -10.0, -10.0)
    assert obs.shape == g5["obstacle"].shape and np.array_equal(obs, g5["obstacle"])
    assert int(np.sum(g5["map_data"] == 50)) > 0           # unknown cells count as obstacles (> 20)


@pytest.mark.parametrize("impl", ["np", "c"])
def test_laser_estimation(g5, impl):
    f = on.laser_estimation if impl == "np" else co.laser_estimation
    for n, key in ((120, "vscan120"), (360, "vscan360")):
        inc = (3.14159 - -3.14159) / (n - 1)
        for k in range(g5["poses"].shape[0]):
            got = f(g5["obstacle"], g5["poses"][k], -3.14159, inc, n)
            if impl == "np":
                assert np.array_equal(got, g5[key][k]), (n, k)    # same math.hypot / atan2: bit-exact
            else:   # C hypot() and CPython's math.hypot may differ in the last bit; the bins may not
                assert np.max(np.abs(got - g5[key][k])) < 1e-12 and np.array_equal(got == 100.0, g5[key][k] == 100.0)


def test_map_observation(g5):
    inc = (3.14159 - -3.14159) / 119
    for k in range(g5["obs_T"].shape[0]):
        src = on.laser_to_numpy(g5["obs_ranges"][k], -3.14159, 3.14159)
        T = on.map_observation(g5["obs_wall"], g5["obs_xest"][k], src, -3.14159, 3.14159, inc)
        assert np.max(np.abs(T - g5["obs_T"][k])) < FTOL


# ------------------------------------------------------------------ w12-mapping-online (G6)
@pytest.fixture(scope="module")
def g6():
    from conftest import load_golden
    return load_golden("g6_mapping_online.npz")


def test_occupied_rule_tables():
    assert on.occupied_rule() == [1001]
    assert on.occupied_rule(0.01, 4.0, 10.0) == [1001, 601, 201]
    import ctypes as C
    tab = (C.c_int * 8)()
    L = co.lib()
    L.orc_occupied_rule.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_int]
    assert L.orc_occupied_rule(0.01, 4.0, 10.0, tab, 8) == 3 and list(tab[:3]) == [1001, 601, 201]
    assert L.orc_occupied_rule(0.01, 20.0, 10.0, tab, 8) == 1 and tab[0] == 1001


def test_mapping_online_replay_golden(g6):
    """The float restatement with hit_inc=4 follows the reference's +4 variant bit for bit
    (same order of adds), and the canonical integer rule gives the same map."""
    m = on.Mapping(200, 200, 0.1, hit_inc=4.0)
    snaps = {int(k): i for i, k in enumerate(g6["snap_steps"])}
    for k in range(g6["ox"].shape[0]):
        pmap = m.update(g6["ox"][k], g6["oy"][k], g6["centres"][k, 0], g6["centres"][k, 1])
        if k in snaps:
            assert np.array_equal(pmap.astype(np.int8), g6["pmap_snaps"][snaps[k]]), k
            assert np.array_equal(m.pmap_from_counts(), g6["pmap_snaps"][snaps[k]]), k
    assert np.array_equal(m.datamap, g6["datamap"])


def test_mapping_online_stress_golden(g6):
    m = on.Mapping(200, 200, 0.1, hit_inc=4.0)
    off = np.concatenate([[0], np.cumsum(g6["stress_len"])])
    cx, cy = g6["stress_centre"]
    for k in range(len(off) - 1):
        m.update(g6["stress_ox"][off[k]:off[k + 1]], g6["stress_oy"][off[k]:off[k + 1]], cx, cy)
    assert np.array_equal(m.pmap.astype(np.int8), g6["stress_pmap"])
    assert np.array_equal(m.datamap, g6["stress_datamap"])
    can = m.pmap_from_counts()
    sens = m.order_sensitive_cells()
    assert np.array_equal(can[~sens], g6["stress_pmap"][~sens])
    # every level of the rule is exercised: occupied by passes alone with 0, 1 and 2 hits
    for h, t in enumerate(on.occupied_rule(0.01, 4.0, 10.0)):
        sel = m.hit_cnt == h
        assert np.any(sel & (m.pass_cnt >= t)) and np.any(sel & (m.pass_cnt < t) & (m.pass_cnt > 0)), h
    assert np.any(m.hit_cnt >= 3)


def test_mapping_online_boundary_golden(g6):
    """On a threshold the reference's answer depends on arrival order; the canonical rule is
    the hits-first answer and flags exactly those cells as order-sensitive."""
    X = tuple(int(v) for v in g6["boundary_cell"])
    far, at = (np.array([5.05]), np.array([0.05])), (np.array([3.05]), np.array([0.05]))
    seen_split = False
    for h, p, hits_first, ref_pmap, ref_data in g6["boundary_cases"]:
        h, p = int(h), int(p)
        m = on.Mapping(200, 200, 0.1, hit_inc=4.0)
        seq = [at] * h + [far] * p if hits_first else [far] * p + [at] * h
        for ox, oy in seq:
            m.update(ox, oy, 0.05, 0.05)
        assert (m.hit_cnt[X], m.pass_cnt[X]) == (h, p)
        assert m.pmap[X] == ref_pmap and m.datamap[X] == ref_data       # float path == reference, either order
        can = m.pmap_from_counts()[X]
        if hits_first:
            assert can == ref_pmap
        elif can != ref_pmap:
            seen_split = True
            assert m.order_sensitive_cells()[X]
    assert seen_split        # (2 hits, 200 passes) really is order-dependent in the reference
