"""GPU parity at the STATED sizes of BASELINE.json configs[2], [3] (per-GPU share) and [4],
through the C ABI, against the C oracle (oracle/checks.py), plus the regression tests of
round-1 review findings (host-pointer replay with the "pipeline" option, the LDS-window
sizing that once overran).  Run on the MI355X box with ``-m gpu``."""
import numpy as np
import pytest

from conftest import pkg
from oracle import c_oracle as co
from oracle import checks

pytestmark = pytest.mark.gpu
FTOL = 1e-9
AMIN, AMAX = -3.14159, 3.14159


@pytest.fixture(scope="module")
def slam():
    p = pkg()
    p._abi.default_context()
    return p


def _device_replay(slam, rep, points, xw, yw, reso):
    dr = slam.DeviceReplay(rep.ranges, AMIN, AMAX, dtype=points)
    grid = dr.make_grid(1, xw, yw, reso)
    dr.run()
    poses, T, it = dr.results()
    dev = {"poses": poses[0], "T": T[0], "iters": it[0], "visits": grid.visits()}
    dev.update(grid.read(0, want=("pmap", "pass", "hit")))
    return dr, grid, dev


def _assert_replay_parity(par):
    assert par["iters_equal"], par
    assert par["pose_max_abs_err"] < FTOL and par["T_max_abs_err"] < FTOL, par
    assert par["counter_cell_mismatches"] == 0 and par["pmap_cell_mismatches"] == 0 and par["visits_equal"], par


# ------------------------------------------------------------------ configs[4] as stated
def test_config4_dense_1080_f16_2000x2000_tiles(slam, syn):
    """BASELINE configs[4] in its stated form, all at once: 1080 beams, fp16 point buffers,
    2000x2000 @ 0.02 m map, room x2, the tiled ray cast (automatic on a map this large), 48
    scans.  The oracle's scan matcher is fed the fp16-rounded points, its
    map the float64 ones."""
    rep = syn.make_replay(48, 1080, seed=3, room_scale=2.0, stride=5)
    dr, grid, dev = _device_replay(slam, rep, "f16", 2000, 2000, 0.02)
    par = checks.compare_replay(dev, rep.ranges, AMIN, AMAX, 2000, 2000, 0.02, points="f16")
    _assert_replay_parity(par)
    assert dev["visits"] > 47 * 1080 * 100          # the rays really are hundreds of cells long
    # the fp16 storage does change the answer (otherwise this would prove nothing about it)
    dr64 = slam.DeviceReplay(rep.ranges, AMIN, AMAX, dtype="f64")
    dr64.run()
    assert np.max(np.abs(dr64.results()[1][0] - dev["T"])) > 1e-6
    # the wedges (4), the window mode (3) and direct atomics (0) give the same map as the tiles
    for mode in (4, 3, 0):
        dr.ctx.set_option("grid_mode", mode)
        dr.run()
        r = grid.read(0, want=("pass", "hit"))
        assert np.array_equal(r["pass"], dev["pass"]) and np.array_equal(r["hit"], dev["hit"]), mode


def test_config4_dense_batch_of_trajectories_into_their_own_maps(slam, syn):
    """configs[4]'s shape in batches: L = 3 different 1080-beam trajectories per slam_replay_dev call, each into its OWN
    2000 x 2000 @ 0.02 m map through the direction wedges (round 5: the wedges take a map per trajectory; until then a
    grid_of_traj sent a large map to the window kernel's scattered atomics).  Every trajectory against the oracle on that
    trajectory alone (fp16 point buffers for the matcher, float64 points for the map: SURVEY.md 7.3-5), wedges chosen
    automatically (grid_mode 1) and forced (4); the window (3) and direct atomics (0) give the same maps."""
    L = 3
    reps = [syn.make_replay(24, 1080, seed=3 + l, room_scale=2.0, stride=5) for l in range(L)]
    ranges = np.stack([r.ranges for r in reps])
    dr = slam.DeviceReplay(ranges, AMIN, AMAX, dtype="f16", grid_of_traj=[2, 0, 1])
    grid = dr.make_grid(L, 2000, 2000, 0.02)
    want = {}
    for mode in (1, 4, 3, 0):
        dr.ctx.set_option("grid_mode", mode)
        dr.run()
        poses, T, it = dr.results()
        for l, gi in enumerate([2, 0, 1]):
            r = grid.read(gi, want=("pmap", "pass", "hit"))
            if mode == 1:
                og = checks.metric_grid(2000, 2000, 0.02)
                op, oT, oit, ovis = checks.replay_reference(ranges[l], AMIN, AMAX, og, points="f16")
                assert np.array_equal(it[l], oit) and np.max(np.abs(poses[l] - op)) < FTOL and np.max(np.abs(T[l] - oT)) < FTOL, l
                assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap), l
                want[gi] = (r, ovis)
            else:
                assert np.array_equal(r["pass"], want[gi][0]["pass"]) and np.array_equal(r["hit"], want[gi][0]["hit"]), (mode, l)
        assert grid.visits() == sum(v[1] for v in want.values()), mode
    assert not np.array_equal(want[0][0]["pass"], want[1][0]["pass"])          # the maps really are different trajectories'
    grid.close()
    dr.ctx.close()


# ------------------------------------------------------------------ configs[1] as bench.py runs it since round 5: batches of trajectories
@pytest.mark.parametrize("group", [0, 16])
def test_config1_batch_of_8_trajectories_of_1000_scans(slam, syn, group):
    """What bench.py's default step is made of: L independent 1 000-scan trajectories in ONE
    slam_replay_dev call - ranges [L, 1000, 360], trajectory l into map l of the grid object, one scan-matching launch of
    L x 999 pairs in the full-chip launch shape, L pose chains side by side, one ray-cast launch with 125 / 63 workgroups per
    trajectory.  Here L = 8 DIFFERENT trajectories (seeds 1..8, different start poses): every trajectory's poses,
    transforms, iteration counts, counters, pmap and visits against the C oracle run on that trajectory alone (pairs
    and trajectories are independent: W12m/slam_ekf.py:109-113).  group 0: the library's choice of scans per ray-cast
    workgroup (12), 16: bench.py's."""
    L = 8
    reps = [syn.make_replay(1000, 360, seed=1 + l, stride=5) for l in range(L)]
    ranges = np.stack([r.ranges for r in reps])
    p0 = np.zeros((L, 3))
    p0[1:] = np.random.default_rng(3).normal(0, [0.5, 0.5, 0.7], size=(L - 1, 3))
    dr = slam.DeviceReplay(ranges, AMIN, AMAX, pose0=p0, grid_of_traj=np.arange(L))
    grid = dr.make_grid(L, 400, 400, 0.05)
    dr.ctx.set_option("grid_group", group)
    dr.run()
    poses, T, it = dr.results()
    total = 0
    for l in range(L):
        og = checks.metric_grid(400, 400, 0.05)
        op, oT, oit, ovis = checks.replay_reference(ranges[l], AMIN, AMAX, og, pose0=tuple(p0[l]))
        assert np.array_equal(it[l], oit), l
        assert np.max(np.abs(poses[l] - op)) < FTOL and np.max(np.abs(T[l] - oT)) < FTOL, l
        r = grid.read(l, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap), l
        total += ovis
    assert grid.visits() == total
    assert len({int(v.sum()) for v in it}) > 1           # the trajectories really differ
    # a second pass without reset doubles every counter (integer evidence adds up, mapping.py:42-45)
    c0 = grid.read(0, want=("pass", "hit"))
    dr.run(reset_grid=False)
    c1 = grid.read(0, want=("pass", "hit"))
    assert np.array_equal(c1["pass"], 2 * c0["pass"]) and np.array_equal(c1["hit"], 2 * c0["hit"])
    grid.close()
    dr.ctx.close()


# ------------------------------------------------------------------ configs[3] per-GPU share
def test_config3_share_5000_scan_replay(slam, syn):
    """One rank's share of BASELINE configs[3]: a 5 000-scan trajectory (seed 10 = rank 0's),
    360 beams, 400x400 @ 0.05 m.  4 999 transforms cross the 512-step chunks of the pose
    composition kernel twice."""
    rep = syn.make_replay(5000, 360, seed=10, stride=5)
    _, grid, dev = _device_replay(slam, rep, "f64", 400, 400, 0.05)
    par = checks.compare_replay(dev, rep.ranges, AMIN, AMAX, 400, 400, 0.05)
    _assert_replay_parity(par)
    assert par["scans"] == 4999
    assert int(dev["pass"].sum(dtype=np.int64)) + int(dev["hit"].sum(dtype=np.int64)) == dev["visits"]


# ------------------------------------------------------------------ configs[2] at 10 000 particles
def test_config2_10000_particles(slam, syn):
    """BASELINE configs[2] at its stated size: 10 000 prior hypotheses of one 360-beam scan
    pair, one 400x400 @ 0.05 m map per particle (12.8 GB of counters: the last maps lie beyond
    4 GB offsets), live pmap.  The oracle runs 32 sampled particles incl. the first and last;
    size-independent properties cover the rest."""
    import torch
    P = 10000
    rep = syn.make_replay(2, 360, seed=2, stride=5)
    mats = slam.prior_matrices(syn.particle_priors(P, seed=2))
    pose_prev = np.random.default_rng(4).normal(0, 0.5, size=(P, 3))
    grid = slam.DeviceGrid.metric(P, 400, 400, 0.05)
    grid.live_pmap()
    poses, T, it = slam.particles_host(rep.ranges[0], rep.ranges[1], AMIN, AMAX, mats, pose_prev, grid=grid)
    rng = np.random.default_rng(7)
    sample = sorted(set([0, 1, P // 2, P - 2, P - 1] + rng.integers(0, P, size=27).tolist()))
    par = checks.compare_particles(poses, T, it, lambda p: grid.read(p, want=("pmap", "pass", "hit")), sample,
                                   rep.ranges[0], rep.ranges[1], AMIN, AMAX, mats, pose_prev, 400, 400, 0.05)
    assert par["iters_equal"] and par["pose_max_abs_err"] < FTOL and par["T_max_abs_err"] < FTOL, par
    assert par["counter_cell_mismatches"] == 0 and par["pmap_cell_mismatches"] == 0, par
    # all 10 000: ICP results against the batched oracle (OpenMP), every particle
    tar = np.array(co.laser_to_points(rep.ranges[0], AMIN, AMAX))
    src = np.array(co.laser_to_points(rep.ranges[1], AMIN, AMAX))
    m = mats
    srcs = np.stack([m[:, 0, 0, None] * src[0] + m[:, 0, 1, None] * src[1] + m[:, 0, 2, None],
                     m[:, 1, 0, None] * src[0] + m[:, 1, 1, None] * src[1] + m[:, 1, 2, None]], axis=1)
    oT, oit, _ = co.icp_batch(np.broadcast_to(tar, (P,) + tar.shape), srcs, 30, 0.001)
    assert np.array_equal(it, oit) and np.max(np.abs(T - oT)) < FTOL
    # all 10 000 maps: every ray ends in one hit (or leaves the map), visits == sum of counters,
    # the live pmap equals the rule applied to the counters (on the device, 14 GB: no read-back)
    ps, ht = grid.counters_torch()
    total = int(ps.sum(dtype=torch.int64).item()) + int(ht.sum(dtype=torch.int64).item())
    assert total == grid.visits()
    hits_per_map = ht.reshape(P, -1).sum(dim=1)
    assert int(hits_per_map.max().item()) <= 360 and int(hits_per_map.min().item()) >= 300
    per = 400 * 400
    chunk = 500
    live_ptr = grid.live_pmap()
    for g0 in range(0, P, chunk):
        p_, h_ = ps[g0:g0 + chunk], ht[g0:g0 + chunk]
        want = torch.where((p_ + h_) == 0, 50, torch.where((h_ >= 1) | (p_ >= 1001), 100, 0)).to(torch.int8)
        live = _live_view(torch, live_ptr + g0 * per, (chunk, 400, 400), ps.device)
        assert bool(torch.equal(live, want)), g0
    grid.close()


def _live_view(torch, ptr, shape, device):
    class V:
        pass
    v = V()
    v.__cuda_array_interface__ = {"shape": shape, "typestr": "|i1", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(v, device=device)


# ------------------------------------------------------------------ review findings of round 1
def test_replay_host_with_pipeline_option_and_grid(slam, syn):
    """slam_replay (host pointers) on a context with pipeline=1 and a map: the copies back must
    wait for the compose / map streams (they once raced: stale poses)."""
    rep = syn.make_replay(300, 360, seed=23, stride=5)
    ctx = slam.Context(0)
    ctx.set_option("pipeline", 1)
    og = checks.metric_grid(400, 400, 0.05)
    oposes, oT, oit, ov = co.replay(rep.ranges, AMIN, AMAX, og, threads=8)
    for _ in range(3):
        grid = slam.DeviceGrid.metric(1, 400, 400, 0.05, context=ctx)
        poses, T, it = slam.replay_host(rep.ranges, AMIN, AMAX, grid=grid, context=ctx)
        assert np.array_equal(it, oit) and np.max(np.abs(poses - oposes)) < FTOL and np.max(np.abs(T - oT)) < FTOL
        r = grid.read(0, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap)
        grid.close()
    ctx.set_option("pipeline", 0)
    ctx.close()


def _fan(cx, cy, x_lo, x_hi, y_lo, y_hi, scale, off_x, off_y):
    """World-frame endpoints whose cells lie on the far edges of the cell box [x_lo..x_hi] x
    [y_lo..y_hi] (plus the corners), seen from the cell (cx, cy)."""
    cells = [(x_hi, y) for y in range(y_lo, y_hi + 1)] + [(x, y_hi) for x in range(x_lo, x_hi + 1)]
    cells += [(x_lo, y_lo), (x_lo, y_hi), (x_hi, y_lo)]
    ox = np.array([(c[0] + 0.5) / scale - off_x for c in cells])
    oy = np.array([(c[1] + 0.5) / scale - off_y for c in cells])
    return ox, oy, (cx + 0.5) / scale - off_x, (cy + 0.5) / scale - off_y


@pytest.mark.parametrize("W,H", [(193, 191), (400, 189), (193, 192), (192, 193), (300, 123), (150, 101), (37, 203), (5, 3)])
@pytest.mark.parametrize("y_origin", [100, 101, 102, 103])
@pytest.mark.parametrize("live", [False, True])
def test_window_kernel_subwindow_sizes(slam, W, H, y_origin, live):
    """Deterministic shapes for the LDS-window ray caster's sub-window (grid_mode 3): a bounding
    box that does not fit the 36 864-cell window, wider than 192 cells and with an odd height
    (193x191 once wrote 96 dwords past the window; 400x189 likewise), with and without the live
    pmap (whose quad alignment shifts the window's first row); the last three shapes fit the
    window (with the live pmap: the owner's fast sweep, odd widths and heights).  Counters, pmap and the visit
    count must equal the oracle's, and no internal-error status may be raised."""
    xw, yw, scale, off_x, off_y = 440, 400, 20.0, 11.0, 10.0
    ctx = slam.Context(0)
    ctx.set_option("grid_mode", 3)
    ctx.set_option("grid_group", 1)
    g = slam.DeviceGrid(1, xw, yw, scale, off_x, off_y, context=ctx)
    if live:
        g.live_pmap()
    x_lo, y_lo = 10, y_origin
    ox, oy, cx, cy = _fan(x_lo, y_lo, x_lo, x_lo + W - 1, y_lo, y_lo + H - 1, scale, off_x, off_y)
    og = co.Grid(xw, yw, scale, off_x, off_y)
    for _ in range(2):                                          # twice: the second pass adds to non-zero counters
        g.update_host(ox, oy, cx, cy)
        og.update(ox, oy, cx, cy)
    ctx.check_status()
    r = g.read(0, want=("pmap", "pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt), (W, H, y_origin, live)
    assert np.array_equal(r["pmap"], og.pmap) and g.visits() == og.visits
    bx = np.nonzero(r["pass"].sum(axis=1) + r["hit"].sum(axis=1))[0]
    by = np.nonzero(r["pass"].sum(axis=0) + r["hit"].sum(axis=0))[0]
    assert (bx[-1] - bx[0] + 1, by[-1] - by[0] + 1) == (W, H)   # the box really has the stated shape
    g.close()
    ctx.close()


@pytest.mark.parametrize("case", ["interior_fits", "interior_too_large", "shifted_origins", "axis_rays", "one_half_empty", "odd_rows", "near_map_edge"])
@pytest.mark.parametrize("group", [1, 4, 12])
def test_window_kernel_direction_halves(slam, case, group):
    """The window kernel's two direction halves (grid_mode 3): a group whose bounding box does not
    fit the window gives the window to the rays that run towards larger x and to those that run towards
    smaller x in turn.  Shapes: both halves fit / neither does (sub-rectangle + direct atomics) / scans of
    a group whose origins lie on both sides of the first one's column / rays along the axes (an endpoint in
    the origin's column belongs to the 'larger x' half) / all rays on one side / a map with an odd row
    length (no paired 64-bit flush) / boxes clipped by the map's edge.  Counters, pmap and visit count
    equal the oracle's after two passes."""
    yw = 601 if case == "odd_rows" else 600
    xw, scale, off_x, off_y = 640, 20.0, 16.0, 15.0
    ctx = slam.Context(0)
    ctx.set_option("grid_mode", 3)
    ctx.set_option("grid_group", group)
    g = slam.DeviceGrid(1, xw, yw, scale, off_x, off_y, context=ctx)
    rng = np.random.default_rng(7)
    B = 12
    W, H = (300, 280) if case == "interior_too_large" else (260, 250)
    x_lo, y_lo = (8, 12) if case != "near_map_edge" else (xw - 200, yw - 190)
    scans = []
    for b in range(B):
        cxc, cyc = x_lo + W // 2, y_lo + H // 2
        if case == "shifted_origins":
            cxc += int(rng.integers(-12, 13)); cyc += int(rng.integers(-12, 13))
        if case == "one_half_empty":
            cxc = x_lo
        if case == "axis_rays":
            cells = [(cxc, y_lo), (cxc, y_lo + H - 1), (x_lo, cyc), (x_lo + W - 1, cyc), (cxc, cyc + 1), (cxc + 1, cyc), (cxc - 1, cyc), (cxc, cyc),
                     (x_lo, y_lo), (x_lo + W - 1, y_lo + H - 1), (cxc + 1, y_lo), (cxc - 1, y_lo + H - 1)] * 8
            ox = np.array([(c[0] + 0.5) / scale - off_x for c in cells]); oy = np.array([(c[1] + 0.5) / scale - off_y for c in cells])
            cx, cy = (cxc + 0.5) / scale - off_x, (cyc + 0.5) / scale - off_y
        else:
            x_hi, y_hi = x_lo + W - 1, y_lo + H - 1
            if case == "near_map_edge":
                x_hi += 60; y_hi += 70                                # endpoints beyond the map: their rays leave it
            cells = [(x_hi, y) for y in range(y_lo, y_hi + 1, 3)] + [(x, y_hi) for x in range(x_lo, x_hi + 1, 3)]
            cells += [(x_lo, y) for y in range(y_lo, y_hi + 1, 3)] + [(x, y_lo) for x in range(x_lo, x_hi + 1, 3)]
            cells = cells[b % 3:] + cells[:b % 3]
            ox = np.array([(c[0] + 0.5) / scale - off_x for c in cells]); oy = np.array([(c[1] + 0.5) / scale - off_y for c in cells])
            cx, cy = (cxc + 0.5) / scale - off_x, (cyc + 0.5) / scale - off_y
        scans.append((ox, oy, cx, cy))
    n = min(len(s_[0]) for s_ in scans)
    OX = np.stack([s_[0][:n] for s_ in scans]); OY = np.stack([s_[1][:n] for s_ in scans])
    CX = np.array([s_[2] for s_ in scans]); CY = np.array([s_[3] for s_ in scans])
    og = co.Grid(xw, yw, scale, off_x, off_y)
    for _ in range(2):
        g.update_host(OX, OY, CX, CY)
        for b in range(B):
            og.update(OX[b], OY[b], CX[b], CY[b])
    ctx.check_status()
    r = g.read(0, want=("pmap", "pass", "hit"))
    assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt), (case, group)
    assert np.array_equal(r["pmap"], og.pmap) and g.visits() == og.visits
    assert int(r["pass"].sum()) > 100000
    g.close()
    ctx.close()


def test_dist_replay_sharded_world1_hip_runner(slam, syn):
    """dist.replay_sharded with the HIP runner (the multi-GPU entry of configs[3]) on one
    rank: same poses as the oracle, gathered result == local result."""
    reps = [syn.make_replay(60, 360, seed=10 + i, stride=5) for i in range(3)]
    finals, local, (lo, hi) = slam.dist.replay_sharded(lambda i: reps[i].ranges, 3, AMIN, AMAX, maps=(400, 400, 0.05))
    assert (lo, hi) == (0, 3) and finals.shape == (3, 3) and local.shape == (3, 59, 3)
    maps = slam.dist.replay_sharded.local_maps
    assert maps.shape == (3, 400, 400)
    for i in range(3):
        og = checks.metric_grid(400, 400, 0.05)
        op, _, _, _ = co.replay(reps[i].ranges, AMIN, AMAX, og, threads=8)
        assert np.max(np.abs(local[i] - op)) < FTOL and np.array_equal(finals[i], local[i, -1])
        assert np.array_equal(maps[i], og.pmap), i               # every trajectory's own map
    finals2, local2, _ = slam.dist.replay_sharded(lambda i: reps[i].ranges, 3, AMIN, AMAX)   # without maps: same poses
    assert np.array_equal(local2, local) and slam.dist.replay_sharded.local_maps is None


def test_single_scan_owner_sweep_pass_threshold(slam):
    """The fast owner sweep of a single scan decides pmap from the live pmap's previous value and
    the new pass count: 300 rays along one line per scan, so the line's cells cross the 1001-pass
    threshold during the 4th scan; a cell hit once stays occupied when later only passed through."""
    ctx = slam.Context(0)
    g = slam.DeviceGrid(1, 200, 200, 10.0, 10.0, 10.0, context=ctx)
    g.live_pmap()
    og = co.Grid(200, 200)
    ox, oy = np.full(300, 3.05), np.full(300, 1.55)
    for k in range(5):
        g.update_host(ox, oy, 0.0, 0.0)
        og.update(ox, oy, 0.0, 0.0)
        r = g.read(0, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt), k
        assert np.array_equal(r["pmap"], og.pmap), k
        assert (r["pmap"][110, 105] == 100) == (k >= 3)              # a pass-through cell of the line
    # a shorter ray now ends (hits) in a cell that so far was only passed through, then longer rays
    # pass through that cell again: it stays occupied
    for ex, ey in ((-2.05, 0.0), (-3.05, 0.0), (-3.05, 0.0)):
        g.update_host(np.array([ex]), np.array([ey]), 0.0, 0.0)
        og.update(np.array([ex]), np.array([ey]), 0.0, 0.0)
        r = g.read(0, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt) and np.array_equal(r["hit"], og.hit_cnt) and np.array_equal(r["pmap"], og.pmap)
        assert r["pmap"][79, 100] == 100 and r["pmap"][85, 100] == 0
    assert r["hit"][79, 100] == 1 and r["pass"][79, 100] == 2 and r["hit"][69, 100] == 2
    ctx.check_status()
    g.close()
    ctx.close()


def _ring(x_lo, x_hi, y_lo, y_hi, step=1):
    """Cells on the border of the box, all four sides."""
    cells = [(x, y_lo) for x in range(x_lo, x_hi + 1, step)] + [(x, y_hi) for x in range(x_lo, x_hi + 1, step)]
    cells += [(x_lo, y) for y in range(y_lo, y_hi + 1, step)] + [(x_hi, y) for y in range(y_lo, y_hi + 1, step)]
    return cells


@pytest.mark.parametrize("case", ["fits", "two_strips", "three_strips_many_rays", "leaves_map", "origin_outside", "too_many_rays",
                                  "halves_origin_off_centre", "halves_one_side_too_wide", "halves_rays_leave_map", "halves_axis_rays"])
def test_single_scan_owner_kernel_fans(slam, case):
    """The single-scan owner kernel (one scan into a map with a live pmap: Mapping.update, the
    particle maps): full-circle fans so that every octant, the reference's endpoint swap and
    lines stepping towards -x occur; bounding boxes of one, two and three window strips; one and
    two rays per lane; rays that leave the map; an origin outside the map; more rays than the
    kernel takes (falls back to the group kernel); boxes of two strips that are cut at the origin's column
    instead (both sides fit: "two_strips", an origin off centre, rays that leave the map, rays along the axes
    whose endpoint lies in the origin's column) and one whose wider side does not fit (strips after all).  Two
    passes: the second adds to non-zero counters and meets a pmap that is no longer 50.  Everything must
    equal the oracle."""
    xw, yw, scale, off_x, off_y = 440, 400, 20.0, 11.0, 10.0
    org = (215, 203)
    if case == "fits":
        cells = _ring(150, 290, 120, 280, 2)                       # 141 x 161 cells: one strip, 1 ray per lane
    elif case == "two_strips":
        cells = _ring(60, 330, 90, 300, 3)                         # 271 x 211 -> 224-cell rows: two strips
    elif case == "three_strips_many_rays":
        cells = _ring(20, 420, 30, 370, 2)                         # 401 x 341: three strips, ~740 rays: 2 per lane
    elif case == "leaves_map":
        cells = _ring(-60, 500, -40, 450, 5)                       # every ray ends outside the map
    elif case == "origin_outside":
        org = (-30, 180)
        cells = _ring(40, 300, 100, 320, 4) + [(-50, 181), (-30, 181)]
    elif case == "halves_origin_off_centre":
        org = (150, 230)
        cells = _ring(60, 330, 90, 300, 3)                         # 91 | 181 columns x 224-cell rows: both sides fit
    elif case == "halves_one_side_too_wide":
        org = (75, 140)
        cells = _ring(60, 330, 90, 300, 3)                         # 16 | 256 columns: the right side alone exceeds the window
    elif case == "halves_rays_leave_map":
        org = (215, 203)
        cells = _ring(60, 330, 90, 300, 3) + _ring(-40, 480, 120, 280, 12)   # the box clamps to the map: still two halves
    elif case == "halves_axis_rays":
        org = (200, 200)
        cells = _ring(60, 330, 90, 300, 3) + [(200, y) for y in range(90, 301, 2)] + [(x, 200) for x in range(60, 331, 2)] + [(200, 200)]
    else:
        cells = _ring(100, 400, 50, 350, 1)                        # 1 204 rays > 1 024
    ox = np.array([(c[0] + 0.5) / scale - off_x for c in cells])
    oy = np.array([(c[1] + 0.5) / scale - off_y for c in cells])
    cx, cy = (org[0] + 0.5) / scale - off_x, (org[1] + 0.5) / scale - off_y
    ctx = slam.Context(0)
    g = slam.DeviceGrid(1, xw, yw, scale, off_x, off_y, context=ctx)
    g.live_pmap()
    og = co.Grid(xw, yw, scale, off_x, off_y)
    for k in range(2):
        g.update_host(ox, oy, cx, cy)
        og.update(ox, oy, cx, cy)
        ctx.check_status()
        r = g.read(0, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt), (case, k, int(np.sum(r["pass"] != og.pass_cnt)))
        assert np.array_equal(r["hit"], og.hit_cnt), (case, k)
        assert np.array_equal(r["pmap"], og.pmap), (case, k, int(np.sum(r["pmap"] != og.pmap)))
        assert g.visits() == og.visits, (case, k)
    g.close()
    ctx.close()


@pytest.mark.parametrize("case", ["count_127", "count_128_overflows", "tall_box_two_segments", "box_at_capacity", "box_over_capacity",
                                  "leaves_map", "origin_in_corner", "unaligned_rows", "one_ray", "neighbour_cells_only", "hits_pile_up"])
def test_single_scan_byte_window_kernel_cases(slam, case):
    """The byte-window owner kernel (k_grid_update_owner8: one scan of at most 384 beams into a map with a live pmap)
    and its hand-over to the general kernel: 127 identical rays (the largest count a window byte holds) and 128 (the
    byte-sum check fails: re-do list); a box taller than 256 cells (two 64-quad segments per window row); boxes just
    inside and just beyond the window's capacity; rays that leave the map; an origin in the map's corner cell; rows
    that do not start on a quad; a single ray; rays to the origin's neighbours only (no cell between origin and
    hit); many rays ending in one cell.  Three passes each: later passes meet non-zero counters and a settled pmap."""
    xw, yw, scale, off_x, off_y = 400, 400, 20.0, 10.0, 10.0
    org = (200, 201)
    if case == "count_127":
        cells = [(260, 230)] * 127 + _ring(150, 250, 150, 250, 5)
    elif case == "count_128_overflows":
        cells = [(260, 230)] * 128 + _ring(150, 250, 150, 250, 5)
    elif case == "tall_box_two_segments":
        org = (180, 200)
        cells = _ring(150, 230, 30, 370, 3)                        # 81 x 341 cells: rows of 86 quads
    elif case == "box_at_capacity":
        org = (200, 200)
        cells = _ring(50, 359, 100, 340, 5)                        # 310 x 256 (rows of whole 16-cell pieces, 96 .. 351) = 79 360 window bytes <= 79 808
    elif case == "box_over_capacity":
        org = (200, 200)
        cells = _ring(30, 370, 60, 345, 6)                         # 341 x 304: beyond the window -> general kernel
    elif case == "leaves_map":
        cells = _ring(150, 250, 150, 250, 5) + [(450, 200), (200, -30), (-5, -5)]
    elif case == "origin_in_corner":
        org = (0, 0)
        cells = [(x, 120) for x in range(0, 150, 3)] + [(140, y) for y in range(0, 120, 3)]
    elif case == "unaligned_rows":
        org = (199, 203)
        cells = _ring(151, 249, 153, 251, 3)                       # y range 153 .. 251: rows start one cell past a quad
    elif case == "one_ray":
        cells = [(230, 190)]
    elif case == "neighbour_cells_only":
        cells = [(199, 200), (201, 202), (200, 202), (201, 201), (200, 201)]   # the last one: start == end, empty path
    else:
        cells = [(240, 240)] * 60 + [(240, 241)] * 50 + [(160, 170)] * 3
    ox = np.array([(c[0] + 0.5) / scale - off_x for c in cells])
    oy = np.array([(c[1] + 0.5) / scale - off_y for c in cells])
    cx, cy = (org[0] + 0.5) / scale - off_x, (org[1] + 0.5) / scale - off_y
    assert len(cells) <= 384
    ctx = slam.Context(0)
    g = slam.DeviceGrid(1, xw, yw, scale, off_x, off_y, context=ctx)
    g.live_pmap()
    og = co.Grid(xw, yw, scale, off_x, off_y)
    for k in range(3):
        g.update_host(ox, oy, cx, cy)
        og.update(ox, oy, cx, cy)
        ctx.check_status()
        r = g.read(0, want=("pmap", "pass", "hit"))
        assert np.array_equal(r["pass"], og.pass_cnt), (case, k, int(np.sum(r["pass"] != og.pass_cnt)))
        assert np.array_equal(r["hit"], og.hit_cnt), (case, k)
        assert np.array_equal(r["pmap"], og.pmap), (case, k, int(np.sum(r["pmap"] != og.pmap)))
        assert g.visits() == og.visits, (case, k)
    g.close()
    ctx.close()


@pytest.mark.parametrize("case", ["random_ranges", "inf_and_tiny", "span_270deg", "span_over_one_turn", "quantised_ties",
                                  "near_origin", "n2", "n9", "n63", "nan_ranges", "f16_random", "f32_room"])
def test_scan_window_nearest_neighbour_edge_scans(slam, syn, case):
    """The beam-window nearest-neighbour search of scan targets (nn_polar) must give the exhaustive
    scan's answer whatever the scans look like: unstructured ranges (wide windows: the box search
    takes over), inf / tiny ranges, a 270-degree lidar (no wrap), more than one turn (not a usable
    scan: boxes), quantised ranges (exact ties across beams), queries next to the origin, tiny
    scans, NaN ranges, reduced point storage.  Iteration counts exact, transforms to 1e-9."""
    rng = np.random.default_rng(abs(hash(case)) % 2**31)
    amin, amax, n, scans, points = AMIN, AMAX, 360, 12, "f64"
    if case in ("random_ranges", "f16_random"):
        ranges = rng.uniform(0.1, 30.0, size=(scans, n))
        points = "f16" if case == "f16_random" else "f64"
    elif case == "inf_and_tiny":
        ranges = syn.make_replay(scans, n, seed=5, stride=5).ranges.astype(np.float64)
        ranges[:, ::7] = np.inf
        ranges[:, 3::11] = 0.1
    elif case == "span_270deg":
        amin, amax = -2.356, 2.356
        ranges = syn.make_replay(scans, n, seed=6, stride=5).ranges.astype(np.float64)
    elif case == "span_over_one_turn":
        amin, amax = -4.0, 4.0
        ranges = syn.make_replay(scans, n, seed=7, stride=5).ranges.astype(np.float64)
    elif case == "quantised_ties":
        ranges = np.round(syn.make_replay(scans, n, seed=8, stride=5).ranges.astype(np.float64) * 2.0) / 2.0 + 0.5
    elif case == "near_origin":
        ranges = rng.uniform(0.1, 0.3, size=(scans, n))
    elif case in ("n2", "n9", "n63"):
        n = int(case[1:])
        ranges = syn.make_replay(scans, n, seed=9, stride=5).ranges.astype(np.float64)
    elif case == "nan_ranges":
        ranges = syn.make_replay(scans, n, seed=10, stride=5).ranges.astype(np.float64)
        ranges[2, 50] = np.nan
        ranges[5, 300:304] = np.nan
    else:
        ranges = syn.make_replay(scans, n, seed=11, stride=5).ranges.astype(np.float64)
        points = "f32"
    ranges = ranges.astype(np.float32)
    poses, T, it = slam.replay_host(ranges, amin, amax, dtype=points)
    oposes, oT, oit, _ = checks.replay_reference(ranges, amin, amax, None, points, 30, 1e-3, threads=8)
    assert np.array_equal(it, oit), (case, it, oit)
    assert np.allclose(T, oT.reshape(T.shape), rtol=0, atol=FTOL, equal_nan=True), case
    assert np.allclose(poses, oposes, rtol=0, atol=1e-8, equal_nan=True), case
    # the matcher's stand-alone operator on the same clouds (point buffers: the box search) agrees too
    pts = np.stack([np.array(co.laser_to_points(r, amin, amax)) for r in ranges])
    if points == "f64" and case != "nan_ranges":
        Tb, itb, _ = slam.icp_batch_host(pts[:-1], pts[1:], 30, 1e-3)
        assert np.array_equal(itb, it) and np.max(np.abs(Tb - T)) < FTOL, case
