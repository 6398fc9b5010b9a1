#!/usr/bin/env python3
"""Where a workgroup of the window ray cast (k_grid_update_win) spends its life, for a replay of L trajectories per call:
phase cycles of every workgroup from the -DSLAM_STAMPS diagnostic build of the library (never shipped; csrc/slam_stamps.h).

    bash gpurun_variants/build_variant.sh stamps -DSLAM_STAMPS
    SLAM_HIP_LIB=$PWD/gpurun_variants/libslamhip_stamps.so python tools/grid_stamps.py [L=8] [group=0] [scans=1000]

Prints per configuration: workgroups, launch span on the 100 MHz clock, mean / max lifetime, mean cycles per phase
(scan constants | pass 1 | sort + zero | walk | flush + hits)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401

slam = importlib.import_module("a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd")
A = slam._abi
L_ = A.lib()
L_.slam_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L_.slam_debug_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
NAMES = ["consts", "pass 1", "sort+zero", "walk", "flush+hits"]


def run(L, group, scans, split=None):
    rep = slam.synthetic.make_replay(scans, 360, seed=1, stride=5)
    ranges = np.ascontiguousarray(np.broadcast_to(rep.ranges[None], (L,) + rep.ranges.shape))
    dr = slam.DeviceReplay(ranges, -3.14159, 3.14159, grid_of_traj=np.arange(L) if L > 1 else None)
    dr.make_grid(L, 400, 400, 0.05)
    dr.ctx.set_option("grid_group", group)
    if split is not None:
        dr.ctx.set_option("grid_split", split)
    raw = np.zeros(64, dtype=np.uint32)
    for _ in range(2):
        dr.run()
        dr.ctx.synchronize()
        L_.slam_debug_read(dr.ctx.handle, raw.ctypes.data, 1)
    c = raw[8:].view(np.uint64)
    wgs = int(c[6])
    rec = np.zeros((min(wgs, 32768), 8), dtype=np.uint32)
    L_.slam_debug_records(dr.ctx.handle, rec.ctypes.data, rec.shape[0])
    life = rec[:, 5].astype(np.float64)
    ph = rec[:, :5].astype(np.float64)
    print("L %d, group %d%s: %d workgroups; lifetime mean %.0f k cycles, max %.0f k (%.2f x mean); phases (k cycles, mean): %s"
          % (L, group, "" if split is None else ", split %d" % split, wgs, life.mean() / 1e3, life.max() / 1e3, life.max() / life.mean(),
             ", ".join("%s %.1f" % (n, v / 1e3) for n, v in zip(NAMES, ph.mean(axis=0)))))
    if int(c[12]):
        nb = float(c[12])
        print("     walk per wave-batch of 64 rays (%d batches): set-up %.0f cycles, walk %.0f cycles for %.0f wave-steps (%.1f cycles a step); per workgroup: %.1f batches"
              % (int(c[12]), c[10] / nb, c[11] / nb, c[13] / nb, float(c[11]) / max(float(c[13]), 1.0), nb / max(wgs, 1)))
    order = np.argsort(-life)[:3]
    for k in order:
        print("     longest: block %d traj %d: %s = %.0f k" % (rec[k, 6] & 0xffff, rec[k, 6] >> 16, " ".join("%.0f" % (v / 1e3) for v in ph[k]), life[k] / 1e3))
    dr.ctx.close()


if __name__ == "__main__":
    Ls = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8").split(",")]
    groups = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,8,16").split(",")]
    scans = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    for L in Ls:
        for g in groups:
            run(L, g, scans)
