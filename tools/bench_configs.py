#!/usr/bin/env python3
"""Secondary measurements: the BASELINE.json configurations that are not the bench.py headline
(configs[2] particles, configs[4] dense scan, the 5k-scan share of configs[3]) are run through
`bench.py --config ...` and carry its in-run parity block against the C oracle (--no-check skips
it); the scan-to-map observation (8f-1) and the drop-in call latencies are measured here.  Prints
one JSON line per configuration.  Not part of the driver contract; results are kept under profiles/."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AMIN, AMAX = -3.14159, 3.14159


def timed(fn, steps, warmup, torch):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def via_bench(extra, check=True):
    """The BASELINE.json configurations are bench.py --config modes (replay | particles | dense),
    each with roofline, cpu_baseline and an in-run parity block against the C oracle; this runs
    one of them as a child process and returns its JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + list(extra)
    if not check:
        cmd.append("--no-parity")
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def mapobs(slam, torch, B, steps, warmup):
    """B pose hypotheses of one 360-beam scan against the virtual scan of a
    32k-obstacle map (tests/golden/g5 map): slam_virtual_scan_dev alone and the fused
    slam_map_observation_dev (virtual scan -> points -> ICP)."""
    A = slam._abi
    dev = torch.device("cuda", 0)
    ctx = A.Context(0, torch.cuda.current_stream(dev).cuda_stream)
    g5 = np.load(os.path.join(ROOT, "tests", "golden", "g5_map_observation.npz"))
    obs = g5["obstacle"]
    K, n = obs.shape[1], 360
    rng = np.random.default_rng(4)
    true_pose = np.array([0.5, 0.3, 0.2])
    poses = true_pose + rng.normal(0, [0.1, 0.1, 0.03], size=(B, 3))
    r = slam.synthetic.scans_from_poses(slam.synthetic.World.room(1.0), true_pose[None], n, 5)[0]
    ct, st = A.trig_tables(AMIN, AMAX, n)
    src = np.stack([ct * r.astype(np.float64), st * r.astype(np.float64)])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ox, oy, dp, dsrc, dc, ds = d(obs[0]), d(obs[1]), d(poses), d(src), d(ct), d(st)
    vr = torch.empty((B, n), dtype=torch.float64, device=dev)
    vp = torch.empty((B, 2, n), dtype=torch.float64, device=dev)
    T = torch.empty((B, 9), dtype=torch.float64, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    L = A.lib()
    inc = (AMAX - AMIN) / (n - 1)

    def vs():
        A.check(L.slam_virtual_scan_dev(ctx.handle, ox.data_ptr(), oy.data_ptr(), K, dp.data_ptr(), B, AMIN, inc, n,
                                        vr.data_ptr()))

    def full():
        A.check(L.slam_map_observation_dev(ctx.handle, ox.data_ptr(), oy.data_ptr(), K, dp.data_ptr(), dsrc.data_ptr(),
                                           B, n, 1, dc.data_ptr(), ds.data_ptr(), AMIN, inc, 30, 1e-3, vr.data_ptr(),
                                           vp.data_ptr(), T.data_ptr(), it.data_ptr()))

    dt_vs = timed(vs, steps, warmup, torch)
    dt = timed(full, steps, warmup, torch)
    ctx.check_status()
    return {"config": "8f-1: %d pose hypotheses, 360-beam scan vs virtual scan of a %d-obstacle map" % (B, K),
            "value": B / dt, "unit": "map-observations/s", "ms_per_step": dt * 1e3, "mean_iters": float(it.float().mean()),
            "virtual_scan_ms": dt_vs * 1e3, "virtual_scan_projections_per_s": B * K / dt_vs,
            "virtual_scan_algorithmic_GBps": (B * K * 16 + B * n * 8) / dt_vs / 1e9}


def dropin(slam, torch, steps):
    """Latency of the drop-in classes, one scan per call as a rospy callback makes them (host
    pointers in and out, synchronous): what a user of the reference's API sees."""
    rep = slam.synthetic.make_replay(steps + 2, 360, seed=1, stride=5)
    icp = slam.ICP()
    clouds = [icp.laserToNumpy(rep.message(k)) for k in range(steps + 2)]

    def wall(fn, n):
        fn(0)
        t0 = time.perf_counter()
        for k in range(n):
            fn(k)
        return (time.perf_counter() - t0) / n * 1e3

    t_proc = wall(lambda k: icp.process(clouds[k], clouds[k + 1]), steps)
    m = slam.Mapping.metric(400, 400, 0.05)
    ang = np.linspace(AMIN, AMAX, 360)
    pts = [(np.cos(ang) * rep.ranges[k], np.sin(ang) * rep.ranges[k]) for k in range(steps + 1)]
    t_map = wall(lambda k: m.update(pts[k][0], pts[k][1], 0.0, 0.0), steps)
    node = slam.SLAM_EKF()
    node.laser_count = 4
    msgs = [rep.message(k) for k in range(steps + 2)]

    def cb(k):
        node.laser_count = 4          # process every message
        node.laserCallback(msgs[k + 1])
    t_node = wall(cb, steps)
    t_l2n = wall(lambda k: icp.laserToNumpy(msgs[k]), steps)
    return {"config": "drop-in classes, one 360-beam scan per call (host pointers, synchronous)",
            "ICP.process_ms": t_proc, "Mapping.update_ms": t_map, "SLAM_EKF.laserCallback_ms": t_node,
            "ICP.laserToNumpy_ms": t_l2n, "unit": "ms per call",
            "reference_as_written_ms": {"ICP.process": 1840.0, "Mapping.update": 16.9, "note": "SURVEY.md section 6 (360 beams, this container's CPU)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="particles,dense,long")
    ap.add_argument("--particles", type=int, default=10000)
    ap.add_argument("--grid-group", type=int, default=0)
    ap.add_argument("--grid-mode", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-check", dest="check", action="store_false", help="skip the comparison with the oracle")
    args = ap.parse_args()
    import torch
    slam = importlib.import_module("a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd")
    for w in args.which.split(","):
        if w == "particles":
            out = via_bench(["--config", "particles", "--particles", str(args.particles), "--steps", str(args.steps),
                             "--warmup", str(args.warmup)], args.check)
        elif w == "dropin":
            out = dropin(slam, torch, 200)
        elif w == "mapobs":
            out = mapobs(slam, torch, 4096, args.steps, args.warmup)
        elif w == "dense":
            out = via_bench(["--config", "dense", "--steps", str(args.steps), "--warmup", str(args.warmup),
                             "--grid-group", str(args.grid_group), "--grid-mode", str(args.grid_mode)], args.check)
        elif w == "long":
            out = via_bench(["--config", "replay", "--scans", "5000", "--steps", str(args.steps), "--warmup", str(args.warmup)],
                            args.check)
        else:
            continue
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
