# Does a hipGraph of the replay step (k_icp, k_pose_compose, k_grid_update_win, k_grid_finalize) beat stream-ordered launches?
import os, sys, importlib, ctypes as C, numpy as np, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
slam = importlib.import_module("a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd")
A = slam._abi; L = A.lib()
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
rep = slam.synthetic.make_replay(1000, 360, seed=1, stride=5)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    dr = slam.DeviceReplay(rep.ranges, -3.14159, 3.14159); grid = dr.make_grid(1, 400, 400, 0.05)
    pmap = torch.empty((400 * 400,), dtype=torch.int8, device=dev)
    st = C.c_void_p(s.cuda_stream)
    def step():
        dr.run(reset_grid=True)
        A.check(L.slam_grid_finalize_dev(dr.ctx.handle, grid._h, pmap.data_ptr()))
    for _ in range(10): step()
    dr.ctx.synchronize()
    ref_pmap = pmap.clone(); ref_poses = dr.poses.clone()
    def timeit(fn, K=200):
        for _ in range(20): fn()
        hip.hipStreamSynchronize(st)
        t0 = time.perf_counter()
        for _ in range(K): fn()
        t_host = time.perf_counter() - t0
        hip.hipStreamSynchronize(st)
        return (time.perf_counter() - t0) / K * 1e3, t_host / K * 1e3
    print("stream-ordered: %.4f ms per replay (host enqueue %.4f ms)" % timeit(step))
    g = C.c_void_p(); ge = C.c_void_p()
    rc = hip.hipStreamBeginCapture(st, 2); print("begin capture", rc)
    step()
    rc = hip.hipStreamEndCapture(st, C.byref(g)); print("end capture", rc)
    nn = C.c_size_t(0); hip.hipGraphGetNodes(g, None, C.byref(nn)); print("nodes", nn.value)
    rc = hip.hipGraphInstantiate(C.byref(ge), g, None, None, C.c_size_t(0)); print("instantiate", rc)
    pmap.zero_(); dr.poses.zero_(); torch.cuda.synchronize()
    def gstep(): hip.hipGraphLaunch(ge, st)
    print("graph: %.4f ms per replay (host enqueue %.4f ms)" % timeit(gstep))
    hip.hipStreamSynchronize(st)
    print("same results:", bool((pmap == ref_pmap).all()), bool((dr.poses == ref_poses).all()))
    # ten replays per graph
    g2 = C.c_void_p(); ge2 = C.c_void_p()
    hip.hipStreamBeginCapture(st, 2)
    for _ in range(10): step()
    hip.hipStreamEndCapture(st, C.byref(g2)); hip.hipGraphInstantiate(C.byref(ge2), g2, None, None, C.c_size_t(0))
    def g10(): hip.hipGraphLaunch(ge2, st)
    a, b = timeit(g10, 40)
    print("graph of 10 replays: %.4f ms per replay (host enqueue %.4f ms)" % (a / 10, b / 10))
    print("stream-ordered again: %.4f ms per replay (host enqueue %.4f ms)" % timeit(step))
