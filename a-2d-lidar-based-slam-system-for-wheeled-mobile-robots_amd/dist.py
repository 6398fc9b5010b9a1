"""Multi-GPU sharding of independent scan streams (BASELINE.json configs[3]).

The path shards by construction: ``T_k`` depends only on scans k-1 and k
(W12m/slam_ekf.py:109-113) and trajectories never interact, so every rank (one process
per GPU, ``torch.distributed``; backend "nccl" is RCCL on ROCm) replays its own block of
trajectories with NO data-path collective.  The single exchange is one ``all_gather`` of
the final poses (3 float64 per trajectory) so every rank ends with all of them; an
optional ``all_reduce(SUM)`` merges integer evidence counters when ranks ray-cast into one
shared map (integer sums commute: the merged map is bit-identical for any rank count).

The compute is injected (``runner``): the product passes the HIP ``DeviceReplay``; the
CPU gloo tests pass the oracle, so the sharding / gathering logic is exercised without a GPU.
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(n_units, rank, world):
    """Balanced contiguous block [lo, hi) of ``n_units`` for ``rank`` of ``world``."""
    base, extra = divmod(int(n_units), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (as set by torch.distributed.run).  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def all_gather_poses(local_poses, n_total, device=None):
    """local_poses [L_local, 3] (this rank's block, in ``shard_range`` order) ->
    [n_total, 3] on every rank.  Blocks may differ by one trajectory, so they are padded
    to the largest block for the fixed-size collective."""
    import torch
    import torch.distributed as dist
    local_poses = np.asarray(local_poses, dtype=np.float64).reshape(-1, 3)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_poses.copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = -(-n_total // world)
    pad = np.zeros((cap, 3))
    pad[: local_poses.shape[0]] = local_poses
    t = torch.from_numpy(pad.reshape(-1))
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * cap * 3, dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    out = out.cpu().numpy().reshape(world, cap, 3)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rows.append(out[r, : hi - lo])
    return np.concatenate(rows, axis=0)


def all_reduce_counters(pass_cnt, hit_cnt, device=None):
    """Sum integer evidence counters over ranks (shared-map case)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pass_cnt, hit_cnt
    outs = []
    for a in (pass_cnt, hit_cnt):
        t = torch.from_numpy(np.ascontiguousarray(a).astype(np.int64))
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        outs.append(t.cpu().numpy().astype(np.uint32))
    return outs[0], outs[1]


def all_reduce_grid(grid):
    """Merge the maps that the ranks built from disjoint scans: one in-place RCCL
    ``all_reduce(SUM)`` per counter array, on the device (``grid``: a :class:`DeviceGrid`).

    Stream ordering (no host synchronise): the collectives are ordered on torch's CURRENT stream, the counters are
    produced on the grid's context's streams (its own, non-blocking, unless the context was created on torch's).
    ``slam_stream_order`` makes torch's stream wait for every update the context has enqueued so far - an asynchronous
    ``DeviceReplay.run()`` included - and afterwards makes the context wait for the collectives before its next update
    (``dist.all_reduce`` without ``async_op`` returns with torch's current stream ordered behind RCCL's).
    Runs with any initialised process group, a world of one rank included (the collectives are then the identity; the
    ordering is the same) - tests/test_gpu_rccl.py does that on the one GPU of the test box."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    p, h = grid.counters_torch()
    if not p.is_cuda:                       # (gloo rehearsals hand in host tensors: nothing to order)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        return
    cur = torch.cuda.current_stream(p.device).cuda_stream
    grid._ctx.stream_order(cur, 0)
    dist.all_reduce(p, op=dist.ReduceOp.SUM)
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    grid._ctx.stream_order(cur, 1)


def hip_runner(ranges, angle_min, angle_max, max_iter, tolerance, local_rank, maps=None):
    """Default compute: this rank's trajectories on its GPU (DeviceReplay): scan matching, dead
    reckoning and - with ``maps=(xw, yw, reso)`` - one occupancy map per trajectory, as
    BASELINE.json configs[3] runs them.  Returns poses, or (poses, pmaps [L, xw, yw] int8)."""
    from .replay import DeviceReplay
    ranges = np.asarray(ranges, dtype=np.float32)
    L = ranges.shape[0] if ranges.ndim == 3 else 1
    dr = DeviceReplay(ranges, angle_min, angle_max, max_iter=max_iter, tolerance=tolerance, device=local_rank,
                      grid_of_traj=list(range(L)) if maps else None)
    grid = dr.make_grid(L, int(maps[0]), int(maps[1]), float(maps[2])) if maps else None
    dr.run()
    poses, _T, _it = dr.results()
    if not maps:
        return poses
    pmaps = np.stack([grid.read(g, want=("pmap",))["pmap"] for g in range(L)])
    return poses, pmaps


def replay_sharded(make_ranges, n_traj, angle_min, angle_max, max_iter=30, tolerance=0.001, runner=hip_runner,
                   backend=None, maps=None):
    """Replay ``n_traj`` independent scan streams over all ranks.

    ``make_ranges(i)`` returns trajectory i's float32 ranges [n_scan, n]; each rank only
    materialises its own block.  ``maps=(xw, yw, reso)``: every trajectory also builds its
    occupancy map (they stay on their rank; ``replay_sharded.local_maps`` holds this rank's
    [L_local, xw, yw] int8 afterwards).  Returns (final_poses [n_traj, 3] on every rank,
    local_poses [L_local, n_scan-1, 3], (lo, hi))."""
    rank, world, local = init(backend)
    lo, hi = shard_range(n_traj, rank, world)
    replay_sharded.local_maps = None
    if hi > lo:
        ranges = np.stack([np.asarray(make_ranges(i), dtype=np.float32) for i in range(lo, hi)])
        if maps:
            poses, replay_sharded.local_maps = runner(ranges, angle_min, angle_max, max_iter, tolerance, local, maps=maps)
            poses = np.asarray(poses)
        else:
            poses = np.asarray(runner(ranges, angle_min, angle_max, max_iter, tolerance, local))
        finals = poses[:, -1, :]
    else:
        poses, finals = np.zeros((0, 0, 3)), np.zeros((0, 3))
    device = None
    try:
        import torch
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_backend() == "nccl":
            device = torch.device("cuda", local)
    except Exception:
        pass
    return all_gather_poses(finals, n_traj, device=device), poses, (lo, hi)
