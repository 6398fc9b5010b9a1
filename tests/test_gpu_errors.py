"""Argument validation and error reporting of the C ABI on a live context: every rejected call
returns SLAM_ERR_INVALID with a message, touches nothing, and leaves the context usable."""
import ctypes as C

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    slam = pkg()
    ctx = slam.Context(0)
    return slam, slam._abi, slam._abi.lib(), ctx


def test_invalid_arguments_are_rejected(env):
    slam, A, L, ctx = env
    h = ctx.handle
    d = np.zeros(64)
    f = np.zeros(64, dtype=np.float32)
    i8 = np.zeros(64, dtype=np.int8)
    k = C.c_int(0)
    g = slam.DeviceGrid(1, 8, 8, 10.0, 0.4, 0.4, context=ctx)
    bad = [
        L.slam_icp_batch(h, None, A.ptr(d), 1, 4, 4, A.F64, 0, 0, None, 5, 1e-3, A.ptr(d), None, None),
        L.slam_icp_batch(h, A.ptr(d), A.ptr(d), 0, 4, 4, A.F64, 0, 0, None, 5, 1e-3, A.ptr(d), None, None),
        L.slam_icp_batch(h, A.ptr(d), A.ptr(d), 1, 4, 4, 99, 0, 0, None, 5, 1e-3, A.ptr(d), None, None),
        L.slam_icp_batch(h, A.ptr(d), A.ptr(d), 1, 4, 4, A.F64, 0, 0, None, -1, 1e-3, A.ptr(d), None, None),
        L.slam_nn(h, A.ptr(d), A.ptr(d), 1, 0, 4, A.F64, A.ptr(d), A.ptr(np.zeros(4, dtype=np.int32))),
        L.slam_kabsch2d(h, A.ptr(d), None, 1, 4, A.ptr(d)),
        L.slam_scan_to_points(h, A.ptr(f), A.ptr(d), A.ptr(d), 1, 0, 1, A.F64, A.ptr(d)),
        L.slam_pose_compose(h, A.ptr(d), A.ptr(d), 0, 1, A.ptr(d)),
        L.slam_grid_update(h, g._h, A.ptr(d), A.ptr(d), A.ptr(d), A.ptr(d), 0, 4, None),
        L.slam_grid_update(h, g._h, A.ptr(d), A.ptr(d), A.ptr(d), A.ptr(d), 1, 4, A.ptr(np.array([3], dtype=np.int32))),
        L.slam_grid_update_scans(h, g._h, A.ptr(f), A.ptr(d), A.ptr(d), None, None, 1, 4),
        L.slam_grid_update_scans(h, g._h, A.ptr(f), A.ptr(d), A.ptr(d), A.ptr(d), None, 1, 70000),
        L.slam_grid_read(h, g._h, 5, A.ptr(i8), None, None, None),
        L.slam_grid_live_pmap(h, g._h, None),
        L.slam_grid_counters_dev(h, None, None, None),
        L.slam_map_obstacles(h, A.ptr(i8), 0, 8, 1, 0.1, 0.0, 0.0, A.ptr(d), A.ptr(d), 8, C.byref(k)),
        L.slam_virtual_scan(h, A.ptr(d), A.ptr(d), 4, A.ptr(d), 1, 0.0, 0.0, 8, A.ptr(d)),          # zero angle increment
        L.slam_virtual_scan(h, A.ptr(d), A.ptr(d), 4, A.ptr(d), 0, 0.0, 0.1, 8, A.ptr(d)),
        L.slam_scan_to_points_f64(h, None, A.ptr(d), A.ptr(d), 1, 4, A.ptr(d)),
        L.slam_map_observation(h, A.ptr(d), A.ptr(d), 4, A.ptr(d), None, 1, 8, 1, A.ptr(d), A.ptr(d), 0.0, 0.1, 5, 1e-3, A.ptr(d), None),
        L.slam_replay(h, A.ptr(f), A.ptr(d), A.ptr(d), 1, 1, 8, A.F64, 5, 1e-3, A.ptr(d), None, None, A.ptr(d), None, None),   # n_scan < 2
        L.slam_particles(h, A.ptr(f), A.ptr(d), A.ptr(d), 8, A.F64, None, A.ptr(d), 0, 5, 1e-3, None, A.ptr(d), A.ptr(d), None),
        L.slam_set_option(h, b"no_such_option", 1.0),
        L.slam_set_option(h, b"grid_mode", 7.0),
        L.slam_bresenham_batch(h, None, None, 1, None, None, None, 0),
    ]
    assert all(rc == A.ERR_INVALID for rc in bad), bad
    assert L.slam_last_error()                                   # a message is kept
    hh = C.c_void_p()
    assert L.slam_grid_create(h, 1, 4, 4, 10.0, 0.2, 0.2, 0.01, 1.0, 10.0, C.byref(hh)) == A.ERR_INVALID   # 8 hits cannot exceed thresh
    assert L.slam_grid_create(h, 0, 4, 4, 10.0, 0.2, 0.2, 0.01, 20.0, 10.0, C.byref(hh)) == A.ERR_INVALID
    # nothing above touched the map, and the context still works
    r = g.read(0, want=("pmap", "pass", "hit"))
    assert np.all(r["pmap"] == 50) and r["pass"].sum() == 0 and r["hit"].sum() == 0
    ctx.check_status()
    T, it, _ = slam.icp_batch_host(np.random.default_rng(0).normal(size=(2, 40)), np.random.default_rng(1).normal(size=(2, 40)),
                                   5, 0.0, context=ctx)
    assert it[0] == 5 and np.isfinite(T).all()


def test_python_layer_maps_error_codes(env):
    slam, A, L, ctx = env
    with pytest.raises(slam.SlamError):
        ctx.set_option("grid_group", 1000)
    m = slam.Mapping(200, 200, 0.1, context=ctx)
    with pytest.raises(ValueError):                               # int(nan) in the reference
        m.update([float("nan")], [0.0], 0.0, 0.0)
    with pytest.raises(OverflowError):                            # int(inf)
        m.update([0.0], [float("inf")], 0.0, 0.0)
    m.update([0.5], [0.5], 0.0, 0.0)                              # the sticky flag was cleared
    assert m.pmap.max() == 100


@pytest.mark.parametrize("xw,live", [(200, False), (400, True), (400, False)])
@pytest.mark.parametrize("bad_value", [float("nan"), float("inf")])
def test_map_state_after_a_raising_beam_equals_the_references(env, xw, live, bad_value):
    """mapping.py:29-36 raises at the first beam whose coordinate int() cannot convert, with the
    beams before it applied and none after it.  A single-scan update leaves exactly that map
    (window kernel on the reference's 200 x 200 map, owner kernel with a live pmap)."""
    from oracle import c_oracle as co
    slam, A, L, ctx = env
    rng = np.random.default_rng(11)
    n, i_bad = 90, 37
    scale = 10.0 if xw == 200 else 20.0
    ang = np.linspace(-3.1, 3.1, n)
    r = rng.uniform(2.0, 8.0, size=n)
    ox, oy = r * np.cos(ang), r * np.sin(ang)
    ox[5] = np.inf                                               # skipped beam (only ox is tested, :30): no error
    oy[i_bad] = bad_value                                        # raises here (nan: ValueError, inf: OverflowError)
    oy[i_bad + 9] = np.nan                                       # never reached
    g = slam.DeviceGrid(1, xw, xw, scale, 10.0, 10.0, context=ctx)
    if live:
        g.live_pmap()
    with pytest.raises(ValueError if np.isnan(bad_value) else OverflowError):   # what int() raises in the reference
        g.update_host(ox, oy, 0.25, -0.4)
    og = co.Grid(xw, xw, scale, 10.0, 10.0)
    og.update(ox[:i_bad], oy[:i_bad], 0.25, -0.4)                # the reference's state when the exception leaves update()
    got = g.read(0, want=("pmap", "pass", "hit"))
    assert np.array_equal(got["pass"], og.pass_cnt) and np.array_equal(got["hit"], og.hit_cnt)
    assert np.array_equal(got["pmap"], og.pmap) and g.visits() == og.visits
    g.close()


@pytest.mark.parametrize("grid_mode", [0, 1, 2, 3, 4])
def test_batched_update_stops_each_scan_at_its_first_raising_beam(env, grid_mode):
    """Several scans in ONE call, through every ray-cast path (grid_mode 0: direct atomics, 1 / 3: LDS window, 2: recorded
    walks + tiles, 4: direction wedges): every scan is cast up to its own first beam that int() would raise on, as the
    reference's loop over that scan would have (mapping.py:29-36); the call reports that beam's error.  (The reference's
    caller would not even start the scans after the offending one: those are still applied here - INTEGRATION.md section 5.)"""
    from oracle import c_oracle as co
    slam, A, L, _ = env
    ctx = slam.Context(0)
    ctx.set_option("grid_mode", grid_mode)
    rng = np.random.default_rng(12)
    B, n = 5, 120
    ang = np.linspace(-3.1, 3.1, n)
    r = rng.uniform(2.0, 8.0, size=(B, n))
    cx, cy = rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B)
    ox, oy = cx[:, None] + r * np.cos(ang), cy[:, None] + r * np.sin(ang)
    oy[1, 40] = np.nan
    oy[1, 77] = np.inf                                           # never reached
    ox[3, 3] = np.nan
    ox[4, 100] = np.inf                                          # skipped, not an error (mapping.py:30 tests ox only)
    g = slam.DeviceGrid(1, 400, 400, 20.0, 10.0, 10.0, context=ctx)
    with pytest.raises(ValueError):
        g.update_host(ox, oy, cx, cy)
    og = co.Grid(400, 400, 20.0, 10.0, 10.0)
    for b, stop in ((0, n), (1, 40), (2, n), (3, 3), (4, n)):
        og.update(ox[b, :stop], oy[b, :stop], cx[b], cy[b])
    got = g.read(0, want=("pass", "hit"))
    assert np.array_equal(got["pass"], og.pass_cnt) and np.array_equal(got["hit"], og.hit_cnt) and g.visits() == og.visits
    # the same through the replay form (raw ranges + poses): a NaN range raises at its beam
    g.reset()
    ranges = rng.uniform(2.0, 8.0, size=(B, n)).astype(np.float32)
    ranges[2, 55] = np.nan
    ranges[4, 0] = np.nan
    poses = np.column_stack([cx, cy, rng.uniform(-1, 1, B)])
    m = slam.Mapping.metric(400, 400, 0.05, context=ctx) if hasattr(slam.Mapping, "metric") else None
    if m is not None:
        with pytest.raises(ValueError):
            m.update_scans(ranges, -3.1, 3.1, poses, None)
        og = co.Grid(400, 400, 20.0, 10.0, 10.0)
        from oracle import oracle_np as on
        for b, stop in ((0, n), (1, n), (2, 55), (3, n), (4, 0)):
            obs = on.world_points(poses[b], on.laser_to_numpy(ranges[b], -3.1, 3.1, clip_inf=True))
            og.update(obs[0][:stop], obs[1][:stop], poses[b][0], poses[b][1])
        p, h = m.counters()
        assert np.array_equal(p, og.pass_cnt) and np.array_equal(h, og.hit_cnt)
    g.close()
    ctx.close()
