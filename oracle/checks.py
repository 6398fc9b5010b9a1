"""Parity checks of whole workloads (BASELINE.json configs) against the C oracle.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and by the parity /
cpu_baseline legs of bench.py and tools/bench_configs.py, never by the product.  Every
function takes the DEVICE results as plain NumPy arrays and returns a dict of mismatch
counts / maximum errors, so that a test can assert on it and a benchmark can print it.

Bars (BASELINE.json north_star): iteration counts, counters, cells bit-exact; float64 poses
and transforms to 1e-9 (north-star bar 1e-5).
"""
from __future__ import annotations

import os

import numpy as np

from . import c_oracle as co

NP_POINTS = {"f64": np.float64, "f32": np.float32, "f16": np.float16}


def metric_grid(xw, yw, reso):
    """Oracle grid with the generalised index rule int(S*(x+H)), S = 1/reso, H = xw/(2S)
    (the reference is (S, H) = (10, 10))."""
    s = 1.0 / reso
    s = float(round(s)) if abs(round(s) - s) < 1e-9 else s
    return co.Grid(xw, yw, s, xw / (2.0 * s), yw / (2.0 * s))


def replay_reference(ranges, amin, amax, grid, points="f64", max_iter=30, tol=1e-3, pose0=(0.0, 0.0, 0.0), threads=None):
    """The reference's replay (ICP.process per consecutive pair, publishResult's dead reckoning,
    Mapping.update at the composed pose) on ``ranges`` [n_scan, n]; ``grid`` an oracle Grid or
    None.  ``points``: storage type of the point clouds the scan matcher sees (f32 / f16: the
    matcher gets the float64 points rounded to that type, the map is cast from the float64
    points - SURVEY.md 7.3-5).  Returns (poses, T, iters, visits)."""
    threads = threads or min(os.cpu_count() or 1, 64)
    if points == "f64":
        return co.replay(ranges, amin, amax, grid, max_iter=max_iter, tolerance=tol, pose0=pose0, threads=threads,
                         mt_grid=grid is not None)
    pts64 = np.stack([np.array(co.laser_to_points(r, amin, amax)) for r in ranges])
    pts = pts64.astype(NP_POINTS[points]).astype(np.float64)
    T, it, _ = co.icp_batch(pts[:-1], pts[1:], max_iter, tol)
    poses, sta, visits = np.empty((len(T), 3)), np.array(pose0, dtype=np.float64), 0
    for k in range(len(T)):
        sta = co.compose_pose(sta, T[k])
        poses[k] = sta
        if grid is not None:
            wx, wy = co.world_points(poses[k], pts64[k + 1][0], pts64[k + 1][1])
            before = grid.visits
            grid.update(wx, wy, poses[k][0], poses[k][1])
            visits += grid.visits - before
    return poses, T, it, visits


def replay_reference_results(ranges, amin, amax, xw, yw, reso, points="f64", max_iter=30, tol=1e-3, threads=None, with_grid=True):
    """The reference's answers for one trajectory as a dict (poses, T, iters, visits and - with_grid - the oracle Grid),
    for compare_replay_with: a step of several replays of the same scans solves the reference once."""
    og = metric_grid(xw, yw, reso) if with_grid else None
    poses, T, it, visits = replay_reference(ranges, amin, amax, og, points, max_iter, tol, threads=threads)
    return {"poses": poses, "T": T, "iters": it, "visits": visits, "grid": og}


def compare_replay_with(dev, ref):
    """dev: dict with the device's 'poses' [n-1,3], 'T' [n-1,3,3], 'iters' [n-1] and optionally
    'pass', 'hit', 'pmap' [xw,yw], 'visits'; ref: replay_reference_results(...)."""
    poses, T, it, og = ref["poses"], ref["T"], ref["iters"], ref["grid"]
    out = {"scans": int(len(it)),
           "pose_max_abs_err": float(np.max(np.abs(np.asarray(dev["poses"]) - poses))),
           "T_max_abs_err": float(np.max(np.abs(np.asarray(dev["T"]).reshape(T.shape) - T))),
           "iters_equal": bool(np.array_equal(np.asarray(dev["iters"]).reshape(-1), it))}
    if og is not None and "pass" in dev:
        out["counter_cell_mismatches"] = int(np.sum(dev["pass"] != og.pass_cnt) + np.sum(dev["hit"] != og.hit_cnt))
        if "pmap" in dev:
            out["pmap_cell_mismatches"] = int(np.sum(dev["pmap"] != og.pmap))
        if "visits" in dev:
            out["visits_equal"] = bool(int(dev["visits"]) == int(ref["visits"]))
    return out


def compare_replay(dev, ranges, amin, amax, xw, yw, reso, points="f64", max_iter=30, tol=1e-3, threads=None):
    """One trajectory's device results against the reference solved here (see compare_replay_with)."""
    return compare_replay_with(dev, replay_reference_results(ranges, amin, amax, xw, yw, reso, points, max_iter, tol, threads=threads,
                                                             with_grid="pass" in dev))


def particle_reference(ranges_prev, ranges_cur, amin, amax, prior_mat, pose_prev, xw, yw, reso, max_iter=30, tol=1e-3):
    """One particle hypothesis of BASELINE configs[2] the way the reference's operators would run
    it: ICP.process on the prior-perturbed source, one dead-reckoning step with M = T.prior, and
    Mapping.update of the ORIGINAL scan at the new pose into a fresh map.
    Returns (pose, T, iters, grid)."""
    tar = np.array(co.laser_to_points(ranges_prev, amin, amax))
    src = np.array(co.laser_to_points(ranges_cur, amin, amax))
    m = np.asarray(prior_mat, dtype=np.float64).reshape(2, 3)
    sp = np.stack([m[0, 0] * src[0] + m[0, 1] * src[1] + m[0, 2], m[1, 0] * src[0] + m[1, 1] * src[1] + m[1, 2]])
    T, it, _ = co.icp_process(tar, sp, max_iter, tol)
    M = np.eye(3)
    M[0, 0] = T[0, 0] * m[0, 0] + T[0, 1] * m[1, 0]
    M[1, 0] = T[1, 0] * m[0, 0] + T[1, 1] * m[1, 0]
    M[0, 2] = T[0, 0] * m[0, 2] + T[0, 1] * m[1, 2] + T[0, 2]
    M[1, 2] = T[1, 0] * m[0, 2] + T[1, 1] * m[1, 2] + T[1, 2]
    pose = co.compose_pose(pose_prev, M)
    og = metric_grid(xw, yw, reso)
    ox, oy = co.world_points(pose, src[0], src[1])
    og.update(ox, oy, pose[0], pose[1])
    return pose, T, it, og


def compare_particles(dev_poses, dev_T, dev_iters, read_map, sample, ranges_prev, ranges_cur, amin, amax, prior_mats,
                      pose_prev, xw, yw, reso, max_iter=30, tol=1e-3, steps=1):
    """Check the particles listed in ``sample``.  ``read_map(p)`` returns the device's
    {'pass','hit','pmap'} of particle p's map.  ``steps``: how many times the same update was
    accumulated into the maps (a benchmark repeats the step without resetting: the counters
    are then exact multiples and pmap follows the counters)."""
    out = {"particles_checked": len(sample), "pose_max_abs_err": 0.0, "T_max_abs_err": 0.0, "iters_equal": True,
           "counter_cell_mismatches": 0, "pmap_cell_mismatches": 0}
    thr = co.pass_count_threshold()
    for p in sample:
        pose, T, it, og = particle_reference(ranges_prev, ranges_cur, amin, amax, prior_mats[p], pose_prev[p], xw, yw, reso,
                                             max_iter, tol)
        out["pose_max_abs_err"] = max(out["pose_max_abs_err"], float(np.max(np.abs(dev_poses[p] - pose))))
        out["T_max_abs_err"] = max(out["T_max_abs_err"], float(np.max(np.abs(np.asarray(dev_T[p]).reshape(3, 3) - T))))
        out["iters_equal"] = out["iters_equal"] and int(dev_iters[p]) == int(it)
        m = read_map(p)
        ps, ht = og.pass_cnt.astype(np.int64) * steps, og.hit_cnt.astype(np.int64) * steps
        out["counter_cell_mismatches"] += int(np.sum(m["pass"] != ps) + np.sum(m["hit"] != ht))
        pm = np.where((ps + ht) == 0, 50, np.where((ht >= 1) | (ps >= thr), 100, 0)).astype(np.int8)
        if steps == 1:
            assert np.array_equal(pm, og.pmap)
        out["pmap_cell_mismatches"] += int(np.sum(m["pmap"] != pm))
    return out
