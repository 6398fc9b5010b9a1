#!/usr/bin/env python3
"""What limits overlapping replays?  Times K steps of the replay workload (configs[1]) on L contexts with parts of the step
switched off or varied: no map (scan matching + pose composition only), queries per lane of the scan matcher, one or two
workgroups per group of scans and scans per workgroup of the ray cast.  No events, no parity: wall clock around K steps.

usage (on the GPU box): python tools/overlap_probe.py > gpurun_out/overlap.jsonl        (profiles/r04_overlap_probe.txt)"""
import importlib
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

PKG = "a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd"
slam = importlib.import_module(PKG)
AMIN, AMAX = -3.14159, 3.14159


def run(lanes, with_grid, qpt, split, steps=96, warm=12, finalize=True, scans=1000, group=0):
    rep = slam.synthetic.make_replay(scans, 360, seed=1, stride=5)
    L = slam._abi.lib()
    lns = []
    for _ in range(lanes):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            dr = slam.DeviceReplay(rep.ranges, AMIN, AMAX, max_iter=30, tolerance=1e-3, dtype="f64", device=0)
            grid = dr.make_grid(1, 400, 400, 0.05) if with_grid else None
            pmap = torch.empty((400, 400), dtype=torch.int8, device=dr.dev)
        dr.ctx.set_option("grid_split", split)
        dr.ctx.set_option("grid_group", group)
        dr.ctx.set_option("icp_qpt", qpt)
        lns.append((st, dr, grid, pmap))

    def step(i):
        st, dr, grid, pmap = lns[i % lanes]
        with torch.cuda.stream(st):
            dr.run(reset_grid=True)
            if grid is not None and finalize:
                slam._abi.check(L.slam_grid_finalize_dev(dr.ctx.handle, grid._h, pmap.data_ptr()))
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for st, dr, grid, pmap in lns:
        if grid is not None:
            grid.close()
            dr.grid = None
        dr.ctx.close()
    return dt / steps * 1e3


if __name__ == "__main__":
    for lanes, with_grid, qpt, split in [(4, True, 3, 0), (4, False, 3, 0), (8, False, 3, 0), (2, False, 3, 0), (1, False, 3, 0), (1, False, 2, 0),
                                         (4, True, 2, 0), (4, False, 2, 0), (4, True, 3, 1), (6, True, 3, 0), (8, True, 3, 0)]:
        ms = run(lanes, with_grid, qpt, split)
        print(json.dumps({"lanes": lanes, "grid": with_grid, "qpt": qpt, "split": split, "ms_per_step": ms, "Mscans_s": 0.999 / ms}), flush=True)
    for lanes, group in [(4, 0), (4, 4), (4, 12), (4, 16), (4, 24)]:
        ms = run(lanes, True, 3, 0, group=group)
        print(json.dumps({"lanes": lanes, "grid": True, "qpt": 3, "split": 0, "group": group, "ms_per_step": ms, "Mscans_s": 0.999 / ms}), flush=True)
