"""Batched, device-resident form of the hot path: many scans per launch.

The drop-in classes (``ICP``, ``Mapping``, ``SLAM_EKF``) process one scan per call, as the
reference's callbacks do.  The benchmark configurations of BASELINE.json batch the same
operators: a replay of L scan streams (configs[1], [3], [4]) or a particle batch
(configs[2]).  This module keeps their buffers resident in HBM (torch tensors are used
only as device allocations) and drives ``slam_replay_dev`` / ``slam_icp_batch_dev`` /
``slam_grid_update_dev`` on one stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi


def replay_host(ranges, angle_min, angle_max, grid=None, max_iter=30, tolerance=0.001, pose0=None,
                dtype="f64", grid_of_traj=None, context=None):
    """Host-pointer form (``slam_replay``): ranges float32 [L, n_scan, n] or [n_scan, n]
    -> (poses [L, n_scan-1, 3], T [L, n_scan-1, 3, 3], iters [L, n_scan-1]).  ``grid`` is a
    :class:`DeviceGrid` or ``None``."""
    ctx = context or _abi.default_context()
    r = np.ascontiguousarray(np.asarray(ranges, dtype=np.float32))
    squeeze = r.ndim == 2
    if squeeze:
        r = r[None]
    L, n_scan, n = r.shape
    ct, st = _abi.trig_tables(angle_min, angle_max, n)
    p0 = np.zeros((L, 3)) if pose0 is None else np.ascontiguousarray(np.asarray(pose0, dtype=np.float64).reshape(L, 3))
    poses = np.empty((L, n_scan - 1, 3))
    T = np.empty((L, n_scan - 1, 9))
    it = np.empty((L, n_scan - 1), dtype=np.int32)
    got = None if grid_of_traj is None else np.ascontiguousarray(np.asarray(grid_of_traj, dtype=np.int32))
    _abi.check(_abi.lib().slam_replay(ctx.handle, _abi.ptr(r), _abi.ptr(ct), _abi.ptr(st), L, n_scan, n,
                                      _abi.DTYPES[dtype], int(max_iter), float(tolerance), _abi.ptr(p0),
                                      grid._h if grid is not None else None, _abi.ptr(got), _abi.ptr(poses),
                                      _abi.ptr(T), _abi.ptr(it)))
    T = T.reshape(L, n_scan - 1, 3, 3)
    if squeeze:
        return poses[0], T[0], it[0]
    return poses, T, it


def icp_batch_host(tar, src, max_iter=30, tolerance=0.001, dtype="f64", prior=None, context=None):
    """tar [B,2,M] or [2,M] (shared), src [B,2,N] or [2,N] (shared) ->
    (T [B,3,3], iters [B], mean_err [B]).  ``prior`` [B,2,3]: per-pair affine applied to
    the source first (particle hypotheses)."""
    ctx = context or _abi.default_context()
    code = _abi.DTYPES[dtype]
    npdt = _abi.NP_DTYPES[code]
    tar = np.ascontiguousarray(np.asarray(tar, dtype=npdt))
    src = np.ascontiguousarray(np.asarray(src, dtype=npdt))
    tar_shared, src_shared = tar.ndim == 2, src.ndim == 2
    if prior is not None:
        prior = np.ascontiguousarray(np.asarray(prior, dtype=np.float64).reshape(-1, 6))
    B = (prior.shape[0] if prior is not None else 1) if (tar_shared and src_shared) else (src if tar_shared else tar).shape[0]
    T = np.empty((B, 9))
    it = np.empty(B, dtype=np.int32)
    err = np.empty(B)
    _abi.check(_abi.lib().slam_icp_batch(ctx.handle, _abi.ptr(tar), _abi.ptr(src), B, tar.shape[-1], src.shape[-1],
                                         code, int(tar_shared), int(src_shared), _abi.ptr(prior), int(max_iter),
                                         float(tolerance), _abi.ptr(T), _abi.ptr(it), _abi.ptr(err)))
    return T.reshape(B, 3, 3), it, err


def prior_matrices(priors):
    """(dx, dy, dtheta) perturbations [P,3] -> 2x3 matrices [P,2,3] (rotation by dtheta,
    then translation), the form ``slam_icp_batch`` / ``slam_particles`` take."""
    pr = np.asarray(priors, dtype=np.float64).reshape(-1, 3)
    m = np.zeros((pr.shape[0], 2, 3))
    m[:, 0, 0], m[:, 0, 1], m[:, 0, 2] = np.cos(pr[:, 2]), -np.sin(pr[:, 2]), pr[:, 0]
    m[:, 1, 0], m[:, 1, 1], m[:, 1, 2] = np.sin(pr[:, 2]), np.cos(pr[:, 2]), pr[:, 1]
    return m


def particles_host(ranges_prev, ranges_cur, angle_min, angle_max, prior_mats, pose_prev, grid=None, max_iter=30,
                   tolerance=0.001, dtype="f64", context=None):
    """P pose hypotheses of one scan pair (``slam_particles``, BASELINE.json configs[2]):
    per hypothesis an ICP solve on the prior-perturbed source, one dead-reckoning step from
    ``pose_prev[p]`` and a ray cast of the current scan into map p of ``grid`` (a
    :class:`DeviceGrid` with G >= P maps, or None).  Returns (poses [P,3], T [P,3,3], iters [P])."""
    ctx = context or _abi.default_context()
    r2 = np.ascontiguousarray(np.stack([np.asarray(ranges_prev, dtype=np.float32), np.asarray(ranges_cur, dtype=np.float32)]))
    n = r2.shape[1]
    ct, st = _abi.trig_tables(angle_min, angle_max, n)
    pm = None if prior_mats is None else np.ascontiguousarray(np.asarray(prior_mats, dtype=np.float64).reshape(-1, 6))
    pp = np.ascontiguousarray(np.asarray(pose_prev, dtype=np.float64).reshape(-1, 3))
    P = pp.shape[0]
    if pm is not None and pm.shape[0] != P:
        raise ValueError("prior_mats and pose_prev disagree on the number of hypotheses")
    poses, T, it = np.empty((P, 3)), np.empty((P, 9)), np.empty(P, dtype=np.int32)
    _abi.check(_abi.lib().slam_particles(ctx.handle, _abi.ptr(r2), _abi.ptr(ct), _abi.ptr(st), n, _abi.DTYPES[dtype],
                                         _abi.ptr(pm), _abi.ptr(pp), P, int(max_iter), float(tolerance),
                                         grid._h if grid is not None else None, _abi.ptr(poses), _abi.ptr(T), _abi.ptr(it)))
    return poses, T.reshape(P, 3, 3), it


class DeviceGrid:
    """G occupancy maps resident on the device (``slam_grid_*``)."""

    def __init__(self, G, xw, yw, scale, off_x, off_y, free_inc=0.01, hit_inc=20.0, thresh=10.0, context=None):
        self._ctx = context or _abi.default_context()
        self.G, self.xw, self.yw = int(G), int(xw), int(yw)
        h = C.c_void_p()
        _abi.check(_abi.lib().slam_grid_create(self._ctx.handle, self.G, self.xw, self.yw, float(scale), float(off_x),
                                               float(off_y), float(free_inc), float(hit_inc), float(thresh), C.byref(h)))
        self._h = h

    @classmethod
    def metric(cls, G, xw, yw, reso, **kw):
        """Index rule from the resolution: scale = 1/reso, offsets = half the map width
        (400x400 @ 0.05 -> scale 20, offset 10; the reference is 200x200 @ 0.1 -> 10, 10)."""
        s = 1.0 / reso
        s = float(round(s)) if abs(round(s) - s) < 1e-9 else s
        return cls(G, xw, yw, s, xw / (2.0 * s), yw / (2.0 * s), **kw)

    def close(self):
        if getattr(self, "_h", None) is not None and self._ctx._h is not None:
            _abi.lib().slam_grid_destroy(self._ctx.handle, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _abi.check(_abi.lib().slam_grid_reset(self._ctx.handle, self._h))

    def counters_torch(self):
        """(pass, hit) as torch int32 tensors [G, xw, yw] that ALIAS the device counters (the
        uint32 bits viewed as int32: sums wrap identically) - for checkpoint / restore
        (``.clone()`` / ``.copy_()``) and for ``torch.distributed.all_reduce`` across ranks."""
        import torch
        p, h = C.c_void_p(), C.c_void_p()
        _abi.check(_abi.lib().slam_grid_counters_dev(self._ctx.handle, self._h, C.byref(p), C.byref(h)))

        class _View:                      # minimal __cuda_array_interface__ carrier
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "<i4", "data": (int(ptr), False), "version": 2}
        shape = (self.G, self.xw, self.yw)
        dev = torch.device("cuda", self._ctx.device)
        return (torch.as_tensor(_View(p.value, shape), device=dev), torch.as_tensor(_View(h.value, shape), device=dev))

    def live_pmap(self):
        """Keep ``pmap`` [G, xw, yw] int8 resident and current on the device
        (``slam_grid_live_pmap``); returns its device address."""
        p = C.c_void_p()
        _abi.check(_abi.lib().slam_grid_live_pmap(self._ctx.handle, self._h, C.byref(p)))
        return int(p.value)

    def update_host(self, ox, oy, cx, cy, grid_of_batch=None):
        ox = np.ascontiguousarray(np.asarray(ox, dtype=np.float64))
        oy = np.ascontiguousarray(np.asarray(oy, dtype=np.float64))
        if ox.ndim == 1:
            ox, oy = ox[None], oy[None]
        B, n = ox.shape
        cx = np.ascontiguousarray(np.asarray(cx, dtype=np.float64).reshape(B))
        cy = np.ascontiguousarray(np.asarray(cy, dtype=np.float64).reshape(B))
        gob = None if grid_of_batch is None else np.ascontiguousarray(np.asarray(grid_of_batch, dtype=np.int32))
        _abi.check(_abi.lib().slam_grid_update(self._ctx.handle, self._h, _abi.ptr(ox), _abi.ptr(oy), _abi.ptr(cx),
                                               _abi.ptr(cy), B, n, _abi.ptr(gob)))

    def read(self, g=0, want=("pmap",)):
        out = {}
        if "pmap" in want:
            out["pmap"] = np.empty((self.xw, self.yw), dtype=np.int8)
        if "datamap" in want:
            out["datamap"] = np.empty((self.xw, self.yw), dtype=np.float64)
        if "pass" in want:
            out["pass"] = np.empty((self.xw, self.yw), dtype=np.uint32)
        if "hit" in want:
            out["hit"] = np.empty((self.xw, self.yw), dtype=np.uint32)
        _abi.check(_abi.lib().slam_grid_read(self._ctx.handle, self._h, int(g), _abi.ptr(out.get("pmap")),
                                             _abi.ptr(out.get("datamap")), _abi.ptr(out.get("pass")),
                                             _abi.ptr(out.get("hit"))))
        return out

    def occupancy_grid_data(self, g=0):
        out = np.empty(self.xw * self.yw, dtype=np.int8)
        _abi.check(_abi.lib().slam_grid_occupancy_data(self._ctx.handle, self._h, int(g), _abi.ptr(out)))
        return out

    def visits(self):
        v = C.c_uint64(0)
        _abi.check(_abi.lib().slam_grid_visits(self._ctx.handle, self._h, C.byref(v)))
        return int(v.value)


class DeviceReplay:
    """L scan streams resident in HBM; ``run()`` enqueues one pass of the hot path
    (points -> ICP -> pose composition -> ray casting) with no host traffic.

    torch is used for allocation only; the kernels run on the context's stream, which is
    torch's current stream so torch events / synchronize see them."""

    def __init__(self, ranges, angle_min, angle_max, grid=None, max_iter=30, tolerance=0.001, dtype="f64",
                 pose0=None, grid_of_traj=None, device=0):
        import torch
        if not torch.cuda.is_available():
            raise _abi.SlamError("DeviceReplay needs a GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.torch = torch
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.ctx = _abi.Context(device, stream)
        r = np.ascontiguousarray(np.asarray(ranges, dtype=np.float32))
        if r.ndim == 2:
            r = r[None]
        self.L, self.n_scan, self.n = r.shape
        self.code = _abi.DTYPES[dtype]
        ct, st = _abi.trig_tables(angle_min, angle_max, self.n)
        self.ranges = torch.from_numpy(r).to(self.dev)
        self.cos_t = torch.from_numpy(ct).to(self.dev)
        self.sin_t = torch.from_numpy(st).to(self.dev)
        p0 = np.zeros((self.L, 3)) if pose0 is None else np.asarray(pose0, dtype=np.float64).reshape(self.L, 3)
        self.pose0 = torch.from_numpy(np.ascontiguousarray(p0)).to(self.dev)
        pairs = self.L * (self.n_scan - 1)
        self.poses = torch.empty((self.L, self.n_scan - 1, 3), dtype=torch.float64, device=self.dev)
        self.T = torch.empty((pairs, 9), dtype=torch.float64, device=self.dev)
        self.iters = torch.empty(pairs, dtype=torch.int32, device=self.dev)
        self.grid = grid
        self.got = None
        if grid_of_traj is not None:
            self.got = torch.from_numpy(np.ascontiguousarray(np.asarray(grid_of_traj, dtype=np.int32))).to(self.dev)
        self.max_iter, self.tol = int(max_iter), float(tolerance)
        self._reset_opt = 0
        torch.cuda.synchronize(self.dev)

    def make_grid(self, G, xw, yw, reso, **kw):
        self.grid = DeviceGrid.metric(G, xw, yw, reso, context=self.ctx, **kw)
        return self.grid

    def run(self, reset_grid=True, poses_out=None, T_out=None):
        """One pass.  ``poses_out`` / ``T_out``: optional float64 device tensors
        [L, n_scan-1, 3] / [L*(n_scan-1), 9] to receive the poses / transforms instead of
        ``self.poses`` / ``self.T`` (e.g. slots of a ring: with the context's "pipeline" option
        consecutive replays overlap only if they write different buffers); they become
        ``self.poses`` / ``self.T``."""
        for name, buf in (("poses", poses_out), ("T", T_out)):
            if buf is not None:
                cur = getattr(self, name)
                if tuple(buf.shape) != tuple(cur.shape) or buf.dtype != cur.dtype or not buf.is_contiguous():
                    raise ValueError("%s_out must be a contiguous float64 tensor of shape %r" % (name, tuple(cur.shape)))
                setattr(self, name, buf)
        if self.grid is not None:          # (the reset rides on the scan-matching launch: context option "replay_reset")
            want = 1 if reset_grid else 0
            if want != self._reset_opt:
                self.ctx.set_option("replay_reset", want)
                self._reset_opt = want
        _abi.check(_abi.lib().slam_replay_dev(
            self.ctx.handle, self.ranges.data_ptr(), self.cos_t.data_ptr(), self.sin_t.data_ptr(), self.L, self.n_scan,
            self.n, self.code, self.max_iter, self.tol, self.pose0.data_ptr(),
            self.grid._h if self.grid is not None else None, self.got.data_ptr() if self.got is not None else None,
            None, self.poses.data_ptr(), self.T.data_ptr(), self.iters.data_ptr()))

    @property
    def scans_per_run(self):
        return self.L * (self.n_scan - 1)

    def results(self):
        self.ctx.check_status()
        return (self.poses.cpu().numpy(), self.T.cpu().numpy().reshape(self.L, self.n_scan - 1, 3, 3),
                self.iters.cpu().numpy().reshape(self.L, self.n_scan - 1))
