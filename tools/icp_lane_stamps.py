#!/usr/bin/env python3
"""Lane efficiency of k_icp's beam-window search measured on the device: candidates inside the lanes' own windows against the
candidate slots their waves ran.  Needs the diagnostic build of the library (never shipped):

    gpurun_variants/build_from.sh istamp <csrc> -DSLAM_STAMPS_ICP     (any build of csrc/ with -DSLAM_STAMPS_ICP)
    SLAM_HIP_LIB=.../libslamhip_istamp.so python tools/icp_lane_stamps.py      -> profiles/r04_icp_lane_efficiency.txt

tools/icp_lane_model.py computes the same quantity on the CPU from the kernel's window formula."""
import os, sys, importlib, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
slam = importlib.import_module("a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd")
A = slam._abi; L = A.lib(); L.slam_debug_lanes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for cfg, beams, seed, scale, pts in (("replay", 360, 1, 1.0, "f64"), ("dense", 1080, 3, 2.0, "f16")):
    rep = slam.synthetic.make_replay(1000, beams, seed=seed, room_scale=scale, stride=5)
    for q in (2, 3):
        dr = slam.DeviceReplay(rep.ranges, -3.14159, 3.14159, dtype=pts)
        dr.ctx.set_option("icp_qpt", q)
        buf = np.zeros(4, dtype=np.uint64)
        dr.run(); dr.ctx.synchronize(); L.slam_debug_lanes(dr.ctx.handle, buf.ctypes.data, 1)
        dr.run(); dr.ctx.synchronize(); L.slam_debug_lanes(dr.ctx.handle, buf.ctypes.data, 1)
        it = dr.results()[2]
        print("%s qpt %d: mean iterations %.2f; beam-window search, candidates in the lanes' own windows / candidate slots run: first iteration %.3f (%.2e / %.2e), later iterations %.3f (%.2e / %.2e)"
              % (cfg, q, it.mean(), buf[0] / max(buf[1], 1), buf[0], buf[1], buf[2] / max(buf[3], 1), buf[2], buf[3]))
        dr.ctx.close()
