#!/usr/bin/env python3
"""bench.py - scans/s of the ICP + occupancy-grid hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU (RANK / LOCAL_RANK /
  WORLD_SIZE / MASTER_* from the environment).  Rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1]: a 1k-scan replay, 360 beams, ICP.process +
Mapping.update per scan into a 400x400 @ 0.05 m grid.  "1k scans" are 1000 PROCESSED
scans: every 5th message of a 10 Hz stream, as the reference's callback decimates
(W12m/slam_ekf.py:65-68), so consecutive processed scans are 0.5 s apart.  ICP parameters
are the ones effective in the W12 mapping node, max_iter 30 / tolerance 1e-3
(W12m/icp.py:21-25).  One "step" = one pass of the hot path over that batch, inputs
(float32 ranges) already resident in HBM: map reset -> 999 ICP solves (polar->Cartesian
fused in) -> pose composition -> 999 x 360 rays cast -> pmap finalize.  Unit of `value`: processed
scans per second (one ICP.process + one Mapping.update each), summed over all ranks.
Consecutive steps are independent replays, so they are dealt round-robin to --lanes contexts
(own stream, map and output buffers; default 4) and overlap on the chip; nothing of a step is
skipped or shared, and --check compares the last step of a run with the CPU oracle bit for bit
(cells, counters, iteration counts) / to 1e-9 (poses).  ms_per_step = elapsed / K.

With N > 1 every rank replays its own trajectory (seed 1 + rank; weak scaling, no
data-path collective); the ranks' final poses (3 float64 per replay) are exchanged with RCCL
all_gather, the only exchange BASELINE.json configs[3] has: one collective for all K
replays at the end of the timed region (--gather end, default) or one all_gather per
replay (--gather step).

Extra objects on the JSON line: "roofline" (dominant kernel, timed inside the library with
HIP events carried by every dispatch on its launch stream) and "cpu_baseline" (oracle/slam_oracle.c, the C port of the
reference, on this host's cores; rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd"

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6    # MI355X FP64 vector (half the 157.3 TF FP32 vector rate)
AMIN, AMAX = -3.14159, 3.14159


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scans", type=int, default=1000)
    ap.add_argument("--beams", type=int, default=360)
    ap.add_argument("--grid", type=int, default=400)
    ap.add_argument("--reso", type=float, default=0.05)
    ap.add_argument("--stride", type=int, default=5)
    ap.add_argument("--room-scale", type=float, default=1.0)
    ap.add_argument("--max-iter", type=int, default=30)
    ap.add_argument("--tol", type=float, default=1e-3)
    ap.add_argument("--points", default="f64", choices=["f64", "f32", "f16"],
                    help="storage type of the ICP point buffers (arithmetic is always f64)")
    ap.add_argument("--grid-mode", type=int, default=1, help="1: automatic (LDS window here; tiles on maps much larger than a window), 0: direct global atomics, 2: tiles, 3: window")
    ap.add_argument("--grid-group", type=int, default=-1,
                    help="scans per ray-cast workgroup (0: the library's choice, 8 here; default: 12 when replays overlap "
                         "(measured 0.177 ms per step against 0.187 with 8), else 0)")
    ap.add_argument("--no-timing", action="store_true", help="experiment: no HIP events around the kernels (no roofline)")
    ap.add_argument("--time-lane0-only", action="store_true", help="HIP events on lane 0 only (default: every lane)")
    ap.add_argument("--lanes", type=int, default=4, help="contexts (stream sets) the replays alternate between")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1],
                    help="1: map stage of a replay on a second stream, overlapping the next replay's scan matching")
    ap.add_argument("--gather", default="end", choices=["step", "end", "none"],
                    help="N > 1: all_gather of final poses after every replay (async), once at the end, or never")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check", action="store_true", help="also compare the GPU result with the oracle")
    return ap.parse_args()


def cpu_baseline(rep, args, budget_s):
    """The C port of the reference (oracle/slam_oracle.c, orc_replay_mt: ICP solves and ray
    casting under OpenMP) on this host, same replay, repeated until ~budget_s seconds."""
    from oracle import c_oracle as co
    threads = max(1, min(os.cpu_count() or 1, 64))
    s = round(1.0 / args.reso)
    reps, t_used = 0, 0.0
    while reps < 1 or (t_used < budget_s and reps < 40):
        g = co.Grid(args.grid, args.grid, float(s), args.grid / (2.0 * s), args.grid / (2.0 * s))
        t0 = time.perf_counter()
        co.replay(rep.ranges, AMIN, AMAX, g, max_iter=args.max_iter, tolerance=args.tol, threads=threads, mt_grid=True)
        t_used += time.perf_counter() - t0
        reps += 1
    scans = (rep.ranges.shape[0] - 1) * reps
    return {"value": scans / t_used, "unit": "scans/s", "cores": threads, "kind": "port",
            "sample": "full %d-scan replay x %d repeats (%.1f s), C port of the reference, OpenMP over scan pairs and rays"
                      % (rep.ranges.shape[0], reps, t_used)}


def load_pmc(kernel):
    """Per-launch PMC figures of `kernel` from the committed rocprofv3 --pmc summary, if any."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(kernel, {})
    except Exception:
        return {}


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` (+ source, VALU busy share) from that summary."""
    d = load_pmc(kernel)
    return d.get("hbm_bytes_per_launch"), d.get("source"), d.get("valu_busy_frac")


def main():
    args = parse()
    if args.grid_group < 0:
        args.grid_group = 12 if args.lanes > 1 else 0
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Each lane's stream should own a hardware queue (the HIP runtime multiplexes streams onto
    # GPU_MAX_HW_QUEUES queues, 4 by default): with a process group RCCL adds streams of its own,
    # and two lanes sharing a queue serialise (measured 0.26 instead of 0.18 ms per step).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    dist = None
    # under torch.distributed.run (RANK and MASTER_ADDR set) the collective path is taken even
    # for one rank, so that it can be rehearsed on a one-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    slam = importlib.import_module(PKG)

    rep = slam.synthetic.make_replay(args.scans, args.beams, seed=1 + rank, room_scale=args.room_scale, stride=args.stride)
    # Consecutive replays are independent, so they are overlapped two ways (results are
    # unchanged, see --check):
    #  * --lanes 4 (default): replays alternate between four contexts (own stream, own map and
    #    output buffers; the GPU exposes four hardware queues per process), so one replay's
    #    latency-bound stretches (kernel tails, the ray cast) are filled by the others' work;
    #  * --pipeline 1: inside a context the library runs scan matching, pose composition and
    #    the map stage (reset -> ray cast -> finalize) on three HIP streams, so the map stage of
    #    one replay overlaps the scan matching of the next (4.3 M scans/s with one lane; with
    #    four lanes the extra streams oversubscribe the hardware queues, hence off by default).
    # Every replay writes its poses into its own slot of a ring (T into one of two buffers per
    # lane).  N > 1: the ranks' final poses are all-gathered (RCCL), the only exchange
    # BASELINE.json configs[3] has; nothing is copied per step and no replay stream ever waits
    # for a collective.
    #   --gather end  (default): ONE all_gather of all K replays' final poses at the end of
    #                 the timed region (one larger collective instead of K latency-bound ones);
    #   --gather step: one all_gather per replay (synchronises the lane first).
    class Lane:
        pass

    lanes = []
    for _ in range(max(1, args.lanes)):
        ln = Lane()
        ln.stream = torch.cuda.Stream(device=local) if args.lanes > 1 else torch.cuda.current_stream(local)
        with torch.cuda.stream(ln.stream):
            ln.dr = slam.DeviceReplay(rep.ranges, AMIN, AMAX, max_iter=args.max_iter, tolerance=args.tol,
                                      dtype=args.points, device=local)
            ln.grid = ln.dr.make_grid(1, args.grid, args.grid, args.reso)
            ln.pmap = torch.empty((args.grid, args.grid), dtype=torch.int8, device=ln.dr.dev)
            ln.ring_T = torch.empty((2,) + tuple(ln.dr.T.shape), dtype=torch.float64, device=ln.dr.dev)
        ln.dr.ctx.set_option("grid_mode", args.grid_mode)
        ln.dr.ctx.set_option("grid_group", args.grid_group)
        ln.dr.ctx.set_option("pipeline", args.pipeline)
        ln.count = 0
        lanes.append(ln)
    dr = lanes[0].dr
    slots = args.steps + args.warmup
    ring = torch.empty((slots,) + tuple(dr.poses.shape), dtype=torch.float64, device=dr.dev)
    gathered = torch.empty((slots, world * 3), dtype=torch.float64, device=dr.dev) if use_dist else None
    gathered_all = torch.empty(world * slots * 3, dtype=torch.float64, device=dr.dev) if use_dist else None
    torch.cuda.synchronize()
    done = [0]
    L = slam._abi.lib()

    def step():
        slot = done[0]
        done[0] += 1
        ln = lanes[slot % len(lanes)]
        ln.dr.run(reset_grid=True, poses_out=ring[slot], T_out=ln.ring_T[ln.count & 1])
        ln.count += 1
        slam._abi.check(L.slam_grid_finalize_dev(ln.dr.ctx.handle, ln.grid._h, ln.pmap.data_ptr()))
        if use_dist and args.gather == "step":
            ln.dr.ctx.synchronize()
            dist.all_gather_into_tensor(gathered[slot], ring[slot, 0, -1])

    marks = {}

    def fence():
        for ln in lanes:
            ln.dr.ctx.synchronize()        # joins the lane's compose / map streams
        marks["drained"] = time.perf_counter()
        if use_dist and args.gather == "end":
            dist.all_gather_into_tensor(gathered_all, ring[:, 0, -1, :].contiguous().reshape(-1))
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # HIP events ride on every kernel dispatch of the timed region as its start / stop events
    # (hipExtLaunchKernelGGL inside the library): exact kernel execution times, no queue markers.
    timed_lanes = lanes[:1] if args.time_lane0_only else lanes
    for ln in timed_lanes:
        ln.dr.ctx.timing_enable(not args.no_timing)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    enqueue = time.perf_counter() - t0      # host time to enqueue all steps (launch-bound if ~ elapsed)
    fence()
    elapsed = elapsed_local = time.perf_counter() - t0
    fam = {}
    for ln in timed_lanes:                 # HIP-event times per kernel family
        for k, v in ln.dr.ctx.timing_read().items():
            acc = fam.setdefault(k, [0.0, 0])
            acc[0] += v[0]
            acc[1] += v[1]
        ln.dr.ctx.timing_enable(False)
    last = lanes[(done[0] - 1) % len(lanes)]
    dr, grid, pmap = last.dr, last.grid, last.pmap
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dr.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    scans_per_step = dr.scans_per_run
    visits = grid.visits()                 # in-bounds cell visits of the last step (reset every step)
    poses, T, iters = dr.results()
    total_scans = scans_per_step * world * args.steps
    value = total_scans / elapsed

    if args.no_timing:
        if rank == 0:
            print(json.dumps({"value": value, "ms_per_step": elapsed / args.steps * 1e3, "lanes": args.lanes,
                              "pipeline": args.pipeline, "note": "experiment without HIP-event timing"}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (largest share of the HIP-event time) -----------
    ms = {k: v[0] for k, v in fam.items() if v[1] > 0}
    dom = max(ms, key=ms.get)
    dom_ms, dom_n = fam[dom]
    avg_s = dom_ms / dom_n * 1e-3
    psz = {"f64": 8, "f32": 4, "f16": 2}[args.points]
    # SURVEY 8(d) prices a pair at (n_src+n_tar)*2*s + 72 B for point buffers of s bytes per coordinate.
    # Since polar->Cartesian is fused into k_icp the kernel reads the raw float32 ranges instead
    # (4 B per point) and no point buffer exists: (n_src+n_tar)*4 + 72 B per pair.
    icp_bytes = scans_per_step * ((2 * args.beams) * 4 + 72)
    grid_bytes = 9 * visits                                               # 9 B per in-bounds cell visit
    alg_bytes = {"icp": icp_bytes, "grid": grid_bytes,
                 "compose": scans_per_step * (72 + 24),
                 "finalize": args.grid * args.grid * 9}.get(dom, 0)
    kname = {"icp": "k_icp", "grid": "k_grid_update_replay", "compose": "k_pose_compose",
             "finalize": "k_grid_finalize"}.get(dom, dom)
    if dom == "grid" and args.grid_mode in (1, 3):
        kname = "k_grid_update_win"
    traffic, tsrc, valu_busy = load_traffic(kname)
    achieved = alg_bytes / avg_s / 1e9
    roofline = {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                "valu_busy_frac_pmc": valu_busy,   # SQ_ACTIVE_INST_VALU share of the kernel's SIMD cycles (profiles/)
                "avg_launch_ms": dom_ms / dom_n, "launches": dom_n, "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms_per_step": {k: fam[k][0] / fam[k][1] for k in ms},
                "timed": "start/stop HIP events carried by every dispatch of %d of %d lanes; a duration includes the time the kernel shares the chip with the other lanes' kernels"
                         % (len(timed_lanes), len(lanes))}
    # ICP is VALU-bound, not HBM-bound.  The figure below counts the distance
    # evaluations an EXHAUSTIVE nearest-neighbour scan would make (iters * n_src * n_tar, what the
    # reference does); the pruned search returns the same result evaluating about a fifth of them.
    evals = float(iters.astype(np.int64).sum()) * args.beams * args.beams
    icp_s = fam["icp"][0] / max(fam["icp"][1], 1) * 1e-3
    roofline["icp_work"] = {"exhaustive_equivalent_distance_evals_per_s": evals / icp_s, "mean_iters": float(iters.mean()),
                            "note": "equivalent brute-force rate; f64 VALU peak is %.1f TFLOP/s (~%.1e evals/s at 6 flop each)"
                                    % (F64_VALU_PEAK_TFLOPS, F64_VALU_PEAK_TFLOPS * 1e12 / 6)}
    lds_insts = load_pmc("k_icp").get("lds_insts_per_launch")
    if lds_insts:
        # SURVEY.md 8d asks for the ICP's LDS rate next to its VALU rate: an upper bound from the
        # PMC count of wave-level LDS instructions, pricing each as a 64-lane 16-byte read
        roofline["icp_work"]["lds"] = {"wave_instructions_per_launch": lds_insts,
                                       "upper_bound_GBps": lds_insts * 1024.0 / icp_s / 1e9,
                                       "peak_GBps": 256 * 128 * 2.4,      # 256 CUs x 128 B/clk x 2.4 GHz
                                       "note": "most are per-lane ds_read_b128 of target points; box reads are broadcasts"}
    grid_s = fam["grid"][0] / max(fam["grid"][1], 1) * 1e-3
    roofline["grid_atomics"] = {"cell_visits_per_step": visits, "visits_per_s": visits / grid_s if grid_s else None}

    out = {
        "metric": "scans/sec (360-beam ICP + 0.05 m grid update)", "value": value, "unit": "scans/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "host_enqueue_ms_per_step": enqueue / args.steps * 1e3,
        "closing_collectives_ms": (t0 + elapsed_local - marks["drained"]) * 1e3,   # all_gather + barrier after the last replay
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: %d-scan replay (every %dth message of a 10 Hz stream), %d beams, ICP(max_iter=%d, tol=%g) + %dx%d@%.2fm grid"
                               % (args.scans, args.stride, args.beams, args.max_iter, args.tol, args.grid, args.grid, args.reso),
                   "scans_per_step_per_gpu": scans_per_step, "point_buffers": args.points,
                   "point_buffers_note": "storage type of the points the scan matcher sees (arithmetic is float64 either way); "
                                         "configs[1] names fp32: --points f32 runs at the same speed (0.180 vs 0.178 ms per step), "
                                         "f64 is the default because it is bit-for-bit the reference's own arithmetic",
                   "trajectories_per_gpu": 1,
                   "pipeline": args.pipeline, "lanes": args.lanes, "grid_mode": args.grid_mode, "grid_group": args.grid_group,
                   "parallelism": "1 trajectory per GPU" + (", all_gather of final poses (%s)" % args.gather if world > 1 else "")},
        "roofline": roofline,
    }
    if args.check and rank == 0:
        from oracle import c_oracle as co
        s = round(1.0 / args.reso)
        og = co.Grid(args.grid, args.grid, float(s), args.grid / (2.0 * s), args.grid / (2.0 * s))
        if args.points == "f64":
            op, oT, oit, ov = co.replay(rep.ranges, AMIN, AMAX, og, max_iter=args.max_iter, tolerance=args.tol,
                                        threads=os.cpu_count() or 1, mt_grid=True)
        else:
            # reduced point storage: scan matching sees the points rounded to that type, the map
            # is cast from the float64 points (tests/test_gpu_parity.py::test_replay_reduced_storage_vs_oracle)
            npdt = {"f32": np.float32, "f16": np.float16}[args.points]
            pts64 = np.stack([np.array(co.laser_to_points(r, AMIN, AMAX)) for r in rep.ranges])
            pts = pts64.astype(npdt).astype(np.float64)
            oT, oit, _ = co.icp_batch(pts[:-1], pts[1:], args.max_iter, args.tol)
            op, sta, ov = np.empty((len(oT), 3)), [0.0, 0.0, 0.0], 0
            for k in range(len(oT)):
                sta = co.compose_pose(sta, oT[k])
                op[k] = sta
                wx, wy = co.world_points(op[k], pts64[k + 1][0], pts64[k + 1][1])
                og.update(wx, wy, op[k][0], op[k][1])
            ov = og.visits
        cnt = grid.read(0, want=("pass", "hit"))
        out["parity"] = {"pose_max_abs_err": float(np.max(np.abs(poses[0] - op))), "iters_equal": bool(np.array_equal(iters[0], oit)),
                         "pmap_cell_mismatches": int(np.sum(pmap.cpu().numpy() != og.pmap)),
                         "counter_cell_mismatches": int(np.sum(cnt["pass"] != og.pass_cnt) + np.sum(cnt["hit"] != og.hit_cnt)),
                         "visits_equal": bool(visits == ov)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rep, args, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
