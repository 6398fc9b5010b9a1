#!/usr/bin/env python3
"""bench.py - throughput of the ICP + occupancy-grid hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 without WORLD_SIZE in the environment: this process (which has not imported torch or
  touched HIP) starts the N ranks itself as `python -m torch.distributed.run --nnodes=1
  --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`, lets rank 0's JSON
  line through and exits with the children's status.  Launched BY torch.distributed.run (RANK /
  LOCAL_RANK / WORLD_SIZE / MASTER_* set) it is one rank per GPU; WORLD_SIZE must equal --gpus.
  Rank 0 prints ONE JSON line.

--config selects the BASELINE.json configuration (all print the same JSON shape with "roofline",
"cpu_baseline" and an in-run "parity" block against the C oracle):
  replay     (default) configs[1]: a 1k-scan replay, 360 beams, ICP.process + Mapping.update per
             scan into a 400x400 @ 0.05 m grid; a step is --traj = 32 independent replays of that
             trajectory in one slam_replay_dev call (a map each).  With N > 1: configs[3], one
             5k-scan trajectory per GPU (seed 10 + rank; 6 replays of it per step), final poses
             exchanged with one RCCL all_gather.
  particles  configs[2]: 10 000 prior hypotheses of one 360-beam scan pair, one 400x400 @ 0.05 m
             map per particle (maps persist across steps, as in a particle filter).
  dense      configs[4]: 1k-scan replay, 1080 beams, 2000x2000 @ 0.02 m grid, fp16 point buffers.

"1k scans" are 1000 PROCESSED scans: every 5th message of a 10 Hz stream, as the reference's
callback decimates (W12m/slam_ekf.py:65-68).  ICP parameters are the ones effective in the W12
mapping node, max_iter 30 / tolerance 1e-3 (W12m/icp.py:21-25).  One "step" = one pass of the hot
path over that batch, inputs (float32 ranges) already resident in HBM:
  replay / dense: map reset -> n-1 ICP solves (polar->Cartesian fused in) -> pose composition ->
                  (n-1) x beams rays cast -> pmap finalize;
  particles:      P ICP solves on the prior-perturbed scan -> P pose steps -> P x 360 rays cast
                  into P maps with pmap kept current (no reset: the maps accumulate).
Unit of `value`: processed scans per second (one ICP.process + one Mapping.update each; for
particles one per hypothesis), summed over all ranks.  Consecutive steps are independent, so
they are dealt round-robin to --lanes contexts (own stream, maps and output buffers; replay: 2)
that overlap on the chip; nothing of a step is skipped or shared - every trajectory of a step
is matched, composed, ray-cast into its own map and finalized, and the in-run parity block
checks every one of them against the oracle.
ms_per_step = elapsed / K.  "single_stream" repeats the measurement with ONE lane (kernels back
to back: per-kernel times there satisfy kernel time <= step time); "sustained" keeps stepping
the same workload until >= 3 s (the secondary configurations: 0.5 s) have passed.
"other_configs": particles, dense, the replay at the W7 launch file's parameters (10, 0) and the
replay as rounds 1-4 ran it (one trajectory per call on four contexts); the two legs with four
contexts run in child processes of this one (_child_leg).  The CPU baseline runs first.

Extra objects on the JSON line: "roofline" (dominant kernel; durations from HIP events carried
by every dispatch on its launch stream), "cpu_baseline" (oracle/slam_oracle.c, the C port of the
reference, on this host's cores; rank 0, N = 1 only) and "parity" (the last step's results
against the oracle: cells / counters / iteration counts exact, poses to 1e-9).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd"

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# what this chip delivers for mixed read + write traffic (tools/ubench_rmw.hip, profiles/r03_ubench_rmw.txt): streaming copy
# 4.5 TB/s, counters read and written back in place with non-temporal loads and stores 4.75 TB/s (reads alone 6.3, writes alone 4.4)
MIXED_RW_PEAK_GBS = 4750.0
F64_VALU_PEAK_TFLOPS = 78.6    # MI355X FP64 vector (half the 157.3 TF FP32 vector rate)
# f64 VALU issue: 256 CUs x 4 SIMDs, one wave64 f64 instruction per 4 cycles per SIMD at 2.4 GHz
F64_ISSUE_PEAK = 1024 * 2.4e9 / 4.0
SIMD_CYCLES_PER_S = 1024 * 2.4e9
# Issue cost of a wave instruction on one SIMD, cycles, measured with four waves per SIMD (tools/ubench_issue.hip,
# profiles/r04_ubench_issue.txt): every float64 instruction incl. compares and min / max 4.2; DPP moves, conversions, 32-bit
# integer multiplies and selects (v_cndmask_b32) 4.2-4.3; v_rcp_f32 and the like 8.3; float64 rcp / rsq / sqrt twice that; plain 32-bit ones 2.35
ISSUE_COST = {"f64": 4.2, "quarter": 4.25, "trans_f32": 8.3, "trans_f64": 16.6, "simple": 2.35}
# ds_add_u32 without return, 64 lanes, 16 waves per CU, every lane walking a line through a 72 KiB window of 16-bit counters
# (the address pattern of the ray casts' walk): 5.95 adds per cycle per CU = 3.68e12 /s on 256 CUs; random dwords 5.56,
# conflict-free 11.1 (tools/ubench_issue.hip part 2, profiles/r04_ubench_issue.txt)
LDS_ATOMIC_PEAK = 3.68e12
AMIN, AMAX = -3.14159, 3.14159

CONFIGS = {
    # name: scans, beams, grid, reso, room_scale, points, seed, lanes
    # replay: 32 independent replays of the trajectory per slam_replay_dev call on 2 overlapping contexts, 16 scans per ray-cast
    # workgroup (profiles/r05_traj_sweep.txt: one 999-pair launch cannot fill the chip - its scan matcher runs 999 pairs in 0.100 ms
    # alone and in 0.049 ms inside a 16-trajectory launch; 8 / 16 / 32 / 48 trajectories x 2 contexts: 13.5 / 14.6 / 15.0 / 15.2 M
    # scans/s in the driver's 20-step form, 13.9 / 15.0 / 15.3 / 15.3 M sustained; rounds 1-4 ran ONE trajectory per call on 4
    # contexts, kept as other_configs.lanes4_single_trajectory)
    "replay": dict(scans=1000, beams=360, grid=400, reso=0.05, room_scale=1.0, points="f64", seed=1, lanes=2, traj=32, grid_group=16),
    "dense": dict(scans=1000, beams=1080, grid=2000, reso=0.02, room_scale=2.0, points="f16", seed=3, lanes=4),
    "particles": dict(scans=2, beams=360, grid=400, reso=0.05, room_scale=1.0, points="f64", seed=2, lanes=2),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 48; 12 for particles / dense)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="replay", choices=sorted(CONFIGS))
    ap.add_argument("--scans", type=int, default=None, help="processed scans per trajectory (default 1000; 5000 with --gpus > 1)")
    ap.add_argument("--particles", type=int, default=10000)
    ap.add_argument("--particle-chunks", type=int, default=0, help="particles: chunks a batch is cut into (scan matching of chunk k + 1 beside the ray cast of chunk k); 0: the library's choice, 1: off")
    ap.add_argument("--pose-spread", type=float, default=0.0, help="particles: standard deviation of the hypotheses' previous poses (m, m, rad); 0: all at the origin (SURVEY.md 8d cfg3)")
    ap.add_argument("--beams", type=int, default=None)
    ap.add_argument("--grid", type=int, default=None)
    ap.add_argument("--reso", type=float, default=None)
    ap.add_argument("--stride", type=int, default=5)
    ap.add_argument("--room-scale", type=float, default=None)
    ap.add_argument("--max-iter", type=int, default=30)
    ap.add_argument("--tol", type=float, default=1e-3)
    ap.add_argument("--points", default=None, choices=["f64", "f32", "f16"],
                    help="storage type of the ICP point buffers (arithmetic is always f64)")
    ap.add_argument("--grid-mode", type=int, default=1, help="1: automatic (LDS window; direction wedges on maps much larger than a window), 0: direct global atomics, 2: recorded walks + tiles, 3: window, 4: wedges")
    ap.add_argument("--grid-group", type=int, default=-1,
                    help="scans per ray-cast workgroup (0: the library's choice; default: the configuration's - 16 for the batched replay, else 0)")
    ap.add_argument("--grid-split", type=int, default=None, choices=[-1, 0, 1], help="window ray cast: two workgroups per group of scans (default: the library's choice for one lane, off when replays overlap)")
    ap.add_argument("--no-timing", action="store_true", help="experiment: no HIP events around the kernels (no roofline)")
    ap.add_argument("--icp-qpt", type=int, default=None, help="scan-matching queries per lane (default: 3 with several lanes, else the library's choice by batch size)")
    ap.add_argument("--lanes", type=int, default=None, help="contexts (stream sets) the replays alternate between")
    ap.add_argument("--traj", type=int, default=None,
                    help="replay / dense: independent replays of the trajectory per slam_replay_dev call (ranges [L, n_scan, n], a map per "
                         "trajectory); a step is then L replays")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1],
                    help="1: map stage of a replay on a second stream, overlapping the next replay's scan matching")
    ap.add_argument("--gather", default="end", choices=["step", "end", "none"],
                    help="N > 1: all_gather of final poses after every replay (async), once at the end, or never")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of the CPU baseline (it runs FIRST, before any GPU leg)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run comparison with the oracle")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the one-lane repeat of the measurement")
    ap.add_argument("--no-other-configs", action="store_true", help="default config only: skip the brief particles / dense runs behind it")
    ap.add_argument("--sustain-seconds", type=float, default=None,
                    help="length of the sustained continuation behind the K steps (default: 3 s for the headline replay configuration, so that "
                         "a 5-second utilisation sampler sees the GPU busy; 0.5 s for every other measurement)")
    ap.add_argument("--check", action="store_true", help="(kept for compatibility: the parity block is always on)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.scans is None:
        # configs[3] (N > 1) names 5k-scan trajectories, configs[1] / [4] a 1k-scan replay
        args.scans = 5000 if (args.gpus > 1 and args.config == "replay") else cfg["scans"]
    if args.traj is None:
        # about 32 000 scan pairs per scan-matching launch: 32 replays of the 1k-scan trajectory, 6 of a 5k-scan one
        args.traj = max(1, round(cfg["traj"] * 1000 / args.scans)) if "traj" in cfg else 1
    for k in ("beams", "grid", "reso", "room_scale", "points", "lanes"):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    if args.steps is None:
        args.steps = 20 if args.config == "replay" else 12     # (multiples of the lane counts 2 / 2 / 4: no lane runs a step more than another)
    if args.warmup is None:
        # at least one untimed step per lane: a lane's first step allocates its workspaces (hundreds of MB for the dense
        # configuration's ray records) and loads its kernels
        args.warmup = 4 if args.config == "replay" else max(2, args.lanes)
    if args.sustain_seconds is None:
        args.sustain_seconds = 3.0 if (args.config == "replay" and args.gpus == 1) else 0.5
    if args.grid_group < 0:
        # the library's choice (8 scans per ray-cast workgroup for ONE 1 000-scan replay, 12 from two on) unless the configuration names
        # one: 16 for the batched replay - 504 workgroups for 8 trajectories, one round on the chip's 512 slots (8 / 12 / 16 / 20 scans:
        # 0.280 / 0.267 / 0.211 / 0.258 ms per 8 trajectories, profiles/r05_traj_sweep.txt)
        args.grid_group = cfg.get("grid_group", 0) if args.traj > 1 else 0
    return args


def spawn_ranks(args):
    """--gpus N > 1 outside torch.distributed.run: start the ranks as child processes.  Nothing
    in this process has imported torch or initialised HIP; it only waits and relays the status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# --------------------------------------------------------------------------------------
# CPU baselines (the C port of the reference on this host's cores), bounded samples
# --------------------------------------------------------------------------------------
def cpu_threads():
    return max(1, min(os.cpu_count() or 1, 64))


def cpu_baseline_replay(rep, args, budget_s):
    """orc_replay_mt (ICP solves and ray casting under OpenMP) on a bounded sample of the same
    replay, repeated until ~budget_s seconds."""
    from oracle import checks
    threads = cpu_threads()
    # a sample the host finishes in seconds: the whole 360-beam replay, the first scans of a dense one
    n = rep.ranges.shape[0] if args.beams <= 400 else min(rep.ranges.shape[0], 120)
    n = min(n, 1000)
    sample = rep.ranges[:n]
    reps, t_used = 0, 0.0
    while reps < 1 or (t_used < budget_s and reps < 40):
        g = checks.metric_grid(args.grid, args.grid, args.reso)
        t0 = time.perf_counter()
        checks.replay_reference(sample, AMIN, AMAX, g, args.points, args.max_iter, args.tol, threads=threads)
        t_used += time.perf_counter() - t0
        reps += 1
    return {"value": (n - 1) * reps / t_used, "unit": "scans/s", "cores": threads, "kind": "port",
            "sample": "first %d scans of the %d-scan replay x %d repeats (%.1f s), C port of the reference, OpenMP over scan pairs%s"
                      % (n, rep.ranges.shape[0], reps, t_used, " and rays" if args.points == "f64" else "; map cast sequentially")}


class _ParticleInputs:
    """The particle workload's host-side inputs (the CPU baseline runs before any GPU object exists)."""
    def __init__(self, slam, args, rank):
        self.P = args.particles
        self.rep = slam.synthetic.make_replay(2, args.beams, seed=2 + rank, stride=args.stride)
        self.mats = slam.prior_matrices(slam.synthetic.particle_priors(self.P, seed=2 + rank))
        self.pose_prev = np.zeros((self.P, 3)) if not args.pose_spread else np.random.default_rng(4 + rank).normal(0, args.pose_spread, size=(self.P, 3))


def replay_seed(args, rank):
    return (10 + rank) if args.gpus > 1 and args.config == "replay" else CONFIGS[args.config]["seed"] + rank


def cpu_baseline_first(slam, args, rank):
    """The CPU baseline of --config from freshly generated inputs (same seeds as the GPU workload), BEFORE any GPU leg: the
    process then ends on its GPU work instead of twelve seconds of host threads."""
    if args.config == "particles":
        return cpu_baseline_particles(_ParticleInputs(slam, args, rank), args, args.cpu_seconds)
    rep = slam.synthetic.make_replay(args.scans, args.beams, seed=replay_seed(args, rank), room_scale=args.room_scale, stride=args.stride)
    return cpu_baseline_replay(rep, args, args.cpu_seconds)


def cpu_baseline_particles(wl, args, budget_s):
    """Particle hypotheses one by one through the C port (ICP.process on the perturbed source, pose
    step, Mapping.update into a fresh map), a thread per particle (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import checks
    threads = cpu_threads()
    done, t_used = 0, 0.0
    r0, r1 = wl.rep.ranges[0], wl.rep.ranges[1]

    def one(p):
        checks.particle_reference(r0, r1, AMIN, AMAX, wl.mats[p], wl.pose_prev[p], args.grid, args.grid, args.reso,
                                  args.max_iter, args.tol)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        while done < wl.P and (done == 0 or t_used < budget_s):
            batch = range(done, min(wl.P, done + 8 * threads))
            t0 = time.perf_counter()
            list(ex.map(one, batch))
            t_used += time.perf_counter() - t0
            done += len(batch)
    return {"value": done / t_used, "unit": "scans/s", "cores": threads, "kind": "port",
            "sample": "first %d of %d particle hypotheses (%.1f s), C port of the reference, one thread per particle" % (done, wl.P, t_used)}


def source_hash():
    """SHA-256 over the kernel sources of the library in this tree (csrc/*.hip, *.h) with comments and blank lines
    removed, 16 hex digits: tools/summarize_profiles.py stamps profiles/pmc_traffic.json with it, and a roofline built on
    PMC figures of OTHER sources says so (stale_pmc).  (A comment edit does not change the code the counters saw.)"""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, PKG, "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, PKG, "csrc", "*.h"))):
        text = open(f, "r", encoding="utf-8", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)              # block comments
        text = re.sub(r"//[^\n]*", "", text)                            # line comments (no string literal of the sources holds //)
        lines = [" ".join(ln.split()) for ln in text.splitlines()]
        h.update(os.path.basename(f).encode())
        h.update("\n".join(ln for ln in lines if ln).encode())
    return h.hexdigest()[:16]


def load_pmc(config, kernel):
    """Per-launch PMC figures of `kernel` from the committed rocprofv3 --pmc summaries, if any
    (profiles/pmc_traffic.json: {config: {kernel: {...}}}; produced by tools/profile_gpu.sh)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}
    d = d.get(config, d if config == "replay" else {})
    if not isinstance(d, dict):
        return {}
    stamp = d.get("_source_hash")
    stale = None
    if stamp != source_hash():
        stale = "profiles/pmc_traffic.json[%s] was collected on kernel sources %s, this tree is %s: re-run tools/profile_gpu.sh" % (config, stamp, source_hash())
    if "+" in kernel:          # a family of kernels launched back to back: byte and instruction counts add up
        parts = [d.get(k, {}) for k in kernel.split("+")]
        if not all(parts):
            return {}
        out = {"source": parts[0].get("source")}
        for key in ("hbm_bytes_per_launch", "valu_insts_per_launch", "lds_insts_per_launch"):
            out[key] = sum(p.get(key) or 0.0 for p in parts)
        main = max(parts, key=lambda p: p.get("valu_insts_per_launch") or 0.0)          # the family's main kernel
        for key in ("valu_busy_frac", "issue_mix", "issue_mix_hw", "lds_conflict_cycle_share", "valu_insts_per_launch_qpt3"):
            if main.get(key) is not None:
                out[key] = main[key]
    else:
        out = dict(d.get(kernel, {}))
    if out and stale:
        out["stale"] = stale
    return out


def issue_cycles(pmc, insts=None):
    """SIMD issue cycles of one launch: the hardware's instruction-class counters (issue_mix_hw: float64 add / mul / fma,
    float64 and float32 transcendentals, conversions) priced with the measured cost per class; classes without a counter of
    their own - float64 compares and min / max, DPP moves and lane exchanges - enter with their STATIC share of the kernel's
    ISA relative to its float64 arithmetic (issue_mix.static_classes, k_icp only) scaled by the counted float64 arithmetic;
    the rest at the plain 32-bit cost.  Returns (cycles, breakdown) or (None, None)."""
    insts = insts or pmc.get("valu_insts_per_launch")
    hw = pmc.get("issue_mix_hw")
    if not insts or not hw:
        return None, None
    scale = insts / float(pmc.get("valu_insts_per_launch") or insts)          # (another launch shape: same mix, its own count)
    n = {"f64_arith": hw["f64_arith"] * scale, "f64_trans": hw["f64_trans"] * scale, "cvt": hw["cvt"] * scale, "trans_f32": hw["trans_f32"] * scale}
    st = (pmc.get("issue_mix") or {}).get("static_classes")
    if st and st.get("f64_arith"):
        n["f64_cmp_minmax_est"] = n["f64_arith"] * st["f64_cmp_minmax"] / float(st["f64_arith"])
        n["dpp_lane_est"] = n["f64_arith"] * (st["dpp_lane"] + st.get("cndmask", 0)) / float(st["f64_arith"])      # (selects issue at the same quarter rate)
    else:
        n["f64_cmp_minmax_est"] = n["dpp_lane_est"] = 0.0
    rest = max(insts - sum(n.values()), 0.0)
    cyc = ((n["f64_arith"] + n["f64_cmp_minmax_est"]) * ISSUE_COST["f64"] + (n["cvt"] + n["dpp_lane_est"]) * ISSUE_COST["quarter"] +
           n["f64_trans"] * ISSUE_COST["trans_f64"] + n["trans_f32"] * ISSUE_COST["trans_f32"] + rest * ISSUE_COST["simple"])
    n["simple"] = rest
    counted = n["f64_arith"] + n["f64_trans"] + n["cvt"] + n["trans_f32"]
    est = n["f64_cmp_minmax_est"] + n["dpp_lane_est"]
    return cyc, {"instructions": n, "cycles_per_instruction": cyc / insts, "cost_cycles": ISSUE_COST,
                 # which share of the launch's instructions the hardware counted by class, which entered by their static share of
                 # the ISA (priced at the float64 / quarter-rate cost), and the rest at the plain 32-bit cost; the PMC-side check of
                 # the resulting fraction is roofline.valu_busy_frac_pmc (SQ_ACTIVE_INST_VALU x 4 cycles / the launch's SIMD cycles)
                 "counted_by_class_share": counted / insts, "estimated_from_static_isa_share": est / insts, "plain_32bit_share": rest / insts,
                 "cycles_if_estimated_classes_were_plain": (cyc - (n["f64_cmp_minmax_est"] * (ISSUE_COST["f64"] - ISSUE_COST["simple"]) +
                                                                   n["dpp_lane_est"] * (ISSUE_COST["quarter"] - ISSUE_COST["simple"]))) / insts,
                 "counters": hw.get("source"), "costs": "profiles/r04_ubench_issue.txt (tools/ubench_issue.hip)",
                 "estimated": "float64 compares / min / max, DPP / lane exchanges and selects have no hardware counter: static share of the kernel's ISA relative to its float64 arithmetic"}


# --------------------------------------------------------------------------------------
# workloads
# --------------------------------------------------------------------------------------
_LANE_STREAMS = {}


def lane_stream(torch, local, idx, n_lanes):
    """The torch stream of lane idx: one lane runs on torch's current stream; several lanes on side streams that are created ONCE
    per process and shared by every workload that follows (the HIP runtime deals streams to its hardware queues in creation
    order: a process that had made two dozen streams for the earlier legs ran the 4-lane legs behind them a third slower -
    lanes of one workload then shared a queue and serialised)."""
    if n_lanes <= 1:
        return torch.cuda.current_stream(local)
    key = (local, idx)
    if key not in _LANE_STREAMS:
        _LANE_STREAMS[key] = torch.cuda.Stream(device=local)
    return _LANE_STREAMS[key]


class ReplayWorkload:
    """configs[1] / [3] / [4]: per lane a DeviceReplay + map + pmap; a step runs on lane slot % lanes."""
    family_kernels = {"icp": "k_icp", "grid": "k_grid_update_win", "compose": "k_pose_compose", "finalize": "k_grid_finalize"}

    def __init__(self, slam, torch, args, rank, local, n_lanes, slots):
        self.slam, self.torch, self.args = slam, torch, args
        seed = replay_seed(args, rank)
        self.rep = slam.synthetic.make_replay(args.scans, args.beams, seed=seed, room_scale=args.room_scale, stride=args.stride)
        self.seed = seed
        # --traj L: L independent replays of the trajectory per slam_replay_dev call (ranges [L, n_scan, n], trajectory l
        # into map l of the grid object): one scan-matching launch of L x (n_scan - 1) pairs, one ray-cast launch, L chains of
        # pose composition side by side.  Pairs are independent (W12m/slam_ekf.py:109-113), so nothing changes per trajectory.
        self.traj = L = max(1, int(getattr(args, "traj", 1) or 1))
        ranges = self.rep.ranges if L == 1 else np.ascontiguousarray(np.broadcast_to(self.rep.ranges[None], (L,) + self.rep.ranges.shape))

        class Lane:
            pass
        self.lanes = []
        for li in range(max(1, n_lanes)):
            ln = Lane()
            ln.stream = lane_stream(torch, local, li, n_lanes)
            with torch.cuda.stream(ln.stream):
                ln.dr = slam.DeviceReplay(ranges, AMIN, AMAX, max_iter=args.max_iter, tolerance=args.tol,
                                          dtype=args.points, device=local, grid_of_traj=np.arange(L) if L > 1 else None)
                ln.grid = ln.dr.make_grid(L, args.grid, args.grid, args.reso)
                ln.pmap = torch.empty((L, args.grid, args.grid), dtype=torch.int8, device=ln.dr.dev)
                ln.ring_T = torch.empty((2,) + tuple(ln.dr.T.shape), dtype=torch.float64, device=ln.dr.dev)
            ln.dr.ctx.set_option("grid_mode", args.grid_mode)
            ln.dr.ctx.set_option("grid_group", args.grid_group)
            ln.dr.ctx.set_option("pipeline", args.pipeline)
            # several replays share the chip: one workgroup per group of scans (least total work); a lone replay leaves
            # the choice to the library (two, one per direction half: shortest launch)
            ln.dr.ctx.set_option("grid_split", args.grid_split if args.grid_split is not None else (0 if n_lanes > 1 else -1))
            # several replays share the chip: three queries per lane (fewest instructions); a lone
            # replay leaves the choice to the library (two: shortest launch)
            ln.dr.ctx.set_option("icp_qpt", args.icp_qpt if args.icp_qpt is not None else int(os.environ.get("SLAM_BENCH_QPT", 3 if n_lanes > 1 else 0)))
            ln.count = 0
            self.lanes.append(ln)
        self.dev = self.lanes[0].dr.dev
        self.ring = torch.empty((slots,) + tuple(self.lanes[0].dr.poses.shape), dtype=torch.float64, device=self.dev)
        self.done = 0
        self.L = slam._abi.lib()
        self.units_per_step = self.lanes[0].dr.scans_per_run
        if self.tiled():      # maps much larger than a window: two kernels timed as one family (wedges; recorded walks + tiles with --grid-mode 2)
            self.family_kernels = dict(self.family_kernels, grid="k_ray_bits+k_tile_cast" if args.grid_mode == 2 else "k_wedge_sort+k_wedge_cast")

    def tiled(self):
        a = self.args
        return a.grid_mode in (2, 4) or (a.grid_mode == 1 and a.grid * a.grid > 8 * 36864)

    def contexts(self):
        return [ln.dr.ctx for ln in self.lanes]

    def step(self):
        slot = self.done
        self.done += 1
        ln = self.lanes[slot % len(self.lanes)]
        ln.dr.run(reset_grid=True, poses_out=self.ring[slot % self.ring.shape[0]], T_out=ln.ring_T[ln.count & 1])
        ln.count += 1
        self.slam._abi.check(self.L.slam_grid_finalize_dev(ln.dr.ctx.handle, ln.grid._h, ln.pmap.data_ptr()))
        return ln, slot

    def last_lane(self):
        return self.lanes[(self.done - 1) % len(self.lanes)]

    def final_poses(self, slots):
        return self.ring[:slots, 0, -1, :].contiguous().reshape(-1)

    def collect(self):
        ln = self.last_lane()
        poses, T, iters = ln.dr.results()
        self.iters = iters
        self.visits = ln.grid.visits()          # in-bounds cell visits of the last step, all its maps (reset every step)
        pm = ln.pmap.cpu().numpy()
        dev = []                                # one block per trajectory of the step
        for l in range(self.traj):
            d = {"poses": poses[l], "T": T[l], "iters": iters[l], "pmap": pm[l]}
            d.update(ln.grid.read(l, want=("pass", "hit")))
            dev.append(d)
        return dev

    def parity(self, dev):
        """EVERY trajectory of the last step against the oracle (the L trajectories of a step replay the same scans: the
        reference is solved once); visits: the step's total against L x the reference's."""
        from oracle import checks
        a = self.args
        ref = checks.replay_reference_results(self.rep.ranges, AMIN, AMAX, a.grid, a.grid, a.reso, a.points, a.max_iter, a.tol,
                                              threads=cpu_threads())
        out = None
        for d in dev:
            o = checks.compare_replay_with(d, ref)
            if out is None:
                out = o
            else:
                for k, v in o.items():
                    out[k] = (out[k] and v) if isinstance(v, bool) else (max(out[k], v) if k.endswith("_err") else (v if k == "scans" else out[k] + v))
        out["trajectories_checked"] = len(dev)
        out["visits_equal"] = bool(int(self.visits) == len(dev) * int(ref["visits"]))
        return out

    def algorithmic_bytes(self):
        a = self.args
        n = self.units_per_step
        # SURVEY 8(d) prices a pair at (n_src+n_tar)*2*s + 72 B for point buffers of s bytes per coordinate.
        # polar->Cartesian is fused into k_icp: the kernel reads the raw float32 ranges instead (4 B per
        # point) and no point buffer exists: (n_src+n_tar)*4 + 72 B per pair.  Grid: 9 B per in-bounds
        # cell visit (4 B counter read + 4 B write + 1 B pmap).
        return {"icp": n * ((2 * a.beams) * 4 + 72), "grid": 9 * self.visits, "compose": n * (72 + 24),
                "finalize": a.grid * a.grid * 9}

    def workload_name(self):
        a = self.args
        which = "configs[3] share" if a.gpus > 1 and a.config == "replay" else ("configs[4]" if a.config == "dense" else "configs[1]")
        name = ("%s: %d-scan replay (every %dth message of a 10 Hz stream, seed %d), %d beams, ICP(max_iter=%d, tol=%g) + %dx%d@%.2fm grid, %s point buffers"
                % (which, a.scans, a.stride, self.seed, a.beams, a.max_iter, a.tol, a.grid, a.grid, a.reso, a.points))
        if self.traj > 1:
            name += "; %d independent replays of the %d-scan trajectory per step (one slam_replay_dev call: ranges [%d, %d, %d], a map per trajectory)" % (
                self.traj, a.scans, self.traj, a.scans, a.beams)
        return name

    def cpu_baseline(self, budget):
        return cpu_baseline_replay(self.rep, self.args, budget)

    def close(self):
        """Free the maps, contexts and device buffers now (another configuration follows in the same process)."""
        for ln in self.lanes:
            ln.dr.ctx.synchronize()
            ln.grid.close()
            ln.dr.grid = None
            ln.dr.ctx.close()
        self.lanes, self.ring = [], None


class ParticleWorkload:
    """configs[2]: slam_particles_dev on P hypotheses; maps persist (no reset), live pmap.  With several lanes
    consecutive steps (independent: the same scan pair, the same priors) alternate between contexts that each
    own a stream, output buffers and a set of P maps, so that one step's scan matching (vector-issue bound)
    shares the chip with another's ray cast (memory bound)."""
    family_kernels = {"icp": "k_icp", "grid": "k_grid_update_owner8+k_grid_update_owner_redo", "compose": "k_pose_step", "finalize": "k_grid_finalize"}

    def __init__(self, slam, torch, args, rank, local, n_lanes, slots):
        self.slam, self.torch, self.args = slam, torch, args
        A = slam._abi
        self.dev = torch.device("cuda", local)
        torch.cuda.set_device(self.dev)
        P = self.P = args.particles
        self.rep = slam.synthetic.make_replay(2, args.beams, seed=2 + rank, stride=args.stride)
        n = args.beams
        ct, st = A.trig_tables(AMIN, AMAX, n)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.mats = slam.prior_matrices(slam.synthetic.particle_priors(P, seed=2 + rank))
        # the hypotheses differ by their prior perturbation only; every particle
        # starts the step at the same pose (tests/test_gpu_configs.py also runs scattered poses)
        self.pose_prev = np.zeros((P, 3)) if not args.pose_spread else np.random.default_rng(4 + rank).normal(0, args.pose_spread, size=(P, 3))
        self.shared = dict(ranges2=d(self.rep.ranges.astype(np.float32)), cos_t=d(ct), sin_t=d(st), prior=d(self.mats.reshape(P, 6)),
                           pose_prev=d(self.pose_prev))

        class Lane:
            pass
        self.lanes = []
        for li in range(max(1, n_lanes)):
            ln = Lane()
            ln.stream = lane_stream(torch, local, li, n_lanes)
            ln.ctx = A.Context(local, ln.stream.cuda_stream)
            # one lane = kernels back to back (the stand-alone durations the rooflines use): no chunks
            ln.ctx.set_option("particle_chunks", args.particle_chunks if n_lanes > 1 or args.lanes == 1 else 1)
            ln.t = dict(poses=torch.empty((P, 3), dtype=torch.float64, device=self.dev),
                        T=torch.empty((P, 9), dtype=torch.float64, device=self.dev),
                        iters=torch.empty(P, dtype=torch.int32, device=self.dev))
            ln.grid = slam.DeviceGrid.metric(P, args.grid, args.grid, args.reso, context=ln.ctx)
            ln.pmap_ptr = ln.grid.live_pmap()
            ln.done = 0
            self.lanes.append(ln)
        self.ctx, self.t, self.grid = self.lanes[0].ctx, dict(self.shared, **self.lanes[0].t), self.lanes[0].grid
        self.ring = torch.empty((slots, 3), dtype=torch.float64, device=self.dev)
        self.L = A.lib()
        self.done = 0
        self.units_per_step = P

    def contexts(self):
        return [ln.ctx for ln in self.lanes]

    def step(self):
        A, sh, a = self.slam._abi, self.shared, self.args
        ln = self.lanes[self.done % len(self.lanes)]
        t = ln.t
        A.check(self.L.slam_particles_dev(ln.ctx.handle, sh["ranges2"].data_ptr(), sh["cos_t"].data_ptr(), sh["sin_t"].data_ptr(),
                                          a.beams, A.DTYPES[a.points], sh["prior"].data_ptr(), sh["pose_prev"].data_ptr(), self.P,
                                          a.max_iter, a.tol, ln.grid._h, None, t["poses"].data_ptr(), t["T"].data_ptr(),
                                          t["iters"].data_ptr()))
        # (no finalize call: the ray cast keeps the live pmap current, ln.pmap_ptr IS the result)
        ln.done += 1
        self.done += 1
        return None, self.done - 1

    def final_poses(self, slots):
        return self.t["poses"][-1].repeat(slots).contiguous()

    def collect(self):
        for ln in self.lanes:
            ln.ctx.check_status()
        t = self.lanes[0].t
        self.iters = t["iters"].cpu().numpy()
        self.visits = sum(ln.grid.visits() for ln in self.lanes) / max(self.done, 1)     # per step (the maps are never reset)
        return {"poses": t["poses"].cpu().numpy(), "T": t["T"].cpu().numpy(), "iters": self.iters}

    def parity(self, dev):
        from oracle import checks
        a, P = self.args, self.P
        rng = np.random.default_rng(7)
        sample = sorted(set([0, P // 2, P - 1] + rng.integers(0, P, size=21).tolist()))
        out = checks.compare_particles(dev["poses"], dev["T"], dev["iters"], lambda p: self.grid.read(p, want=("pmap", "pass", "hit")),
                                       sample, self.rep.ranges[0], self.rep.ranges[1], AMIN, AMAX, self.mats, self.pose_prev,
                                       a.grid, a.grid, a.reso, a.max_iter, a.tol, steps=self.lanes[0].done)
        out["map_steps_accumulated"] = self.lanes[0].done
        return out

    def algorithmic_bytes(self):
        a = self.args
        return {"icp": self.P * ((2 * a.beams) * 4 + 72 + 48), "grid": 9 * self.visits, "compose": self.P * (72 + 48 + 24 + 24),
                "finalize": self.P * a.grid * a.grid * 9}

    def workload_name(self):
        a = self.args
        return ("configs[2]: %d particle hypotheses (perturbed priors) of one %d-beam scan pair, ICP(max_iter=%d, tol=%g) + one %dx%d@%.2fm map per particle (persistent, live pmap)"
                % (self.P, a.beams, a.max_iter, a.tol, a.grid, a.grid, a.reso))

    def cpu_baseline(self, budget):
        return cpu_baseline_particles(self, self.args, budget)

    def close(self):
        for ln in self.lanes:
            ln.ctx.synchronize()
            ln.grid.close()
            ln.ctx.close()
        self.lanes, self.shared, self.t, self.grid, self.ctx = [], None, None, None, None


class Env:
    """What a measurement needs besides its arguments: the package, torch, the process group."""
    def __init__(self, slam, torch, dist, rank, world, local, use_dist):
        self.slam, self.torch, self.dist = slam, torch, dist
        self.rank, self.world, self.local, self.use_dist = rank, world, local, use_dist


def config_args(base, name):
    """Arguments of another --config with that config's defaults (sizes, lanes, steps), everything else as given."""
    a = argparse.Namespace(**vars(base))
    cfg = CONFIGS[name]
    a.config = name
    for k in ("beams", "grid", "reso", "room_scale", "points", "lanes"):
        setattr(a, k, cfg[k])
    a.traj = cfg.get("traj", 1)
    a.scans = cfg["scans"]
    a.steps, a.warmup = (20, 4) if name == "replay" else (12, max(2, cfg["lanes"]))   # every lane warms up before the timed region
    a.grid_group = cfg.get("grid_group", 0)
    a.icp_qpt = None
    a.sustain_seconds = 0.5
    return a


def measure(args, env, collective=True, want_single=True, want_sustained=True):
    """One workload of --config: warm-up, the timed K steps (barrier + synchronize on both sides, MAX over ranks),
    the sustained continuation, the one-lane repeat.  Returns a dict of raw results; nothing is printed."""
    slam, torch, dist = env.slam, env.torch, env.dist
    use_dist = env.use_dist and collective
    world = env.world if use_dist else 1
    slots = args.steps + args.warmup
    Workload = ParticleWorkload if args.config == "particles" else ReplayWorkload
    wl = Workload(slam, torch, args, env.rank, env.local, args.lanes, slots)
    dev = wl.dev
    gathered = torch.empty((slots, world * 3), dtype=torch.float64, device=dev) if use_dist else None
    gathered_all = torch.empty(world * slots * 3, dtype=torch.float64, device=dev) if use_dist else None
    torch.cuda.synchronize()

    # Every replay writes its poses into its own slot of a ring (T into one of two buffers per
    # lane).  N > 1: the ranks' final poses are all-gathered (RCCL), the only exchange
    # BASELINE.json configs[3] has; nothing is copied per step and no replay stream ever waits
    # for a collective.
    def step(w=wl):
        ln, slot = w.step()
        if use_dist and args.gather == "step" and ln is not None:
            ln.dr.ctx.synchronize()
            dist.all_gather_into_tensor(gathered[slot % slots], w.ring[slot % slots, 0, -1])

    marks = {}

    def fence(w=wl, coll=True):
        for c in w.contexts():
            c.synchronize()                # joins the lane's compose / map streams
        marks["drained"] = time.perf_counter()
        if use_dist and args.gather == "end" and coll:
            dist.all_gather_into_tensor(gathered_all, w.final_poses(slots))
        torch.cuda.synchronize()
        if use_dist and coll:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(w, steps, coll=True, only=None, events=True):
        """Time exactly `steps` steps of workload w; returns (elapsed, enqueue, {family: [ms, launches]}, t0).
        only: the kernel families whose dispatches carry HIP events (None: all of them); events=False: none at all."""
        for c in w.contexts():
            c.timing_enable(events and not args.no_timing, only=only)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(w)
        enqueue = time.perf_counter() - t0     # host time to enqueue all steps (launch-bound if ~ elapsed)
        fence(w, coll)
        elapsed = time.perf_counter() - t0
        fam = {}
        for c in w.contexts():                 # HIP-event times per kernel family
            for k, v in c.timing_read().items():
                acc = fam.setdefault(k, [0.0, 0])
                acc[0] += v[0]
                acc[1] += v[1]
            c.timing_enable(False)
        return elapsed, enqueue, fam, t0

    for _ in range(args.warmup):
        step()
    fence()
    # HIP events ride on every kernel dispatch of the timed region as its start / stop events
    # (hipExtLaunchKernelGGL inside the library): exact kernel execution times, no queue markers.
    # Events on a dispatch cost throughput (four overlapping replays: 3-4 %), so in the contract's K steps only the
    # dominant kernel family carries them when a sustained continuation follows - that one times every family, and
    # the other families' overlapped per-launch durations are taken from it (res["fam_note"]).
    dominant = "icp" if args.config == "replay" else "grid"
    lean = want_sustained and args.sustain_seconds > 0 and not args.no_timing and len(wl.contexts()) > 1
    elapsed, enqueue, fam, t0 = timed(wl, args.steps, only=[dominant] if lean else None)
    elapsed_local = elapsed
    closing_ms = (t0 + elapsed_local - marks["drained"]) * 1e3
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res = {"wl": wl, "elapsed": elapsed, "elapsed_local": elapsed_local, "enqueue": enqueue, "fam": fam, "closing_ms": closing_ms,
           "n_ranks": world, "dev_results": wl.collect(), "single": None, "sustained": None, "fam_note": None,
           "timing_mask": [dominant] if lean else ("all" if not args.no_timing else []), "instrumented": None, "single_pipelined": None}
    if args.no_timing:
        return res
    if lean:
        # the same K steps once more with events on EVERY family's dispatches (how rounds 1-2 measured `value`): the
        # like-for-like figure for comparisons across rounds
        e_all, _, _, _ = timed(wl, args.steps, coll=False)
        res["instrumented"] = {"value": wl.units_per_step * args.steps / e_all, "ms_per_step": e_all / args.steps * 1e3, "timing_mask": "all",
                               "note": "the K steps repeated with HIP events on the dispatches of every kernel family; per-GPU figure, no collective inside"}
        # ... and with none: what the library does for a caller that does not ask for kernel times
        e_none, _, _, _ = timed(wl, args.steps, coll=False, events=False)
        res["instrumented"]["no_events"] = {"value": wl.units_per_step * args.steps / e_none, "ms_per_step": e_none / args.steps * 1e3}

    # ---- sustained: the same workload stepped on past the contract's K steps until >= --sustain-seconds have
    #      passed, with the kernels' HIP-event times kept (the driver's K steps last a few milliseconds: all lanes
    #      start in the scan-matching phase and end in the map phase, which the longer run averages out) ----------
    if want_sustained and args.sustain_seconds > 0:
        per = elapsed / args.steps
        more = int(min(max(args.sustain_seconds / per, 1), 200000))
        more -= more % max(len(wl.contexts()), 1) if more > len(wl.contexts()) else 0     # no lane runs a step more than another
        ts = time.perf_counter()
        dt, _, fam_s, _ = timed(wl, more, coll=False)
        if lean:
            # launches per step of every family from the continuation; the families not timed in the K steps enter with
            # their per-launch duration from there and the launch count the K steps had
            for k, v in fam_s.items():
                if v[1] > 0 and not fam.get(k, [0.0, 0])[1]:
                    n_l = int(round(v[1] * args.steps / float(more)))
                    if n_l > 0:
                        fam[k] = [v[0] / v[1] * n_l, n_l]
            res["fam_note"] = ("HIP events in the timed K steps on the dominant family (%s) only; the other families' overlapped durations are those of "
                               "the sustained continuation (same workload, same lanes), their launch counts those of the K steps" % dominant)
        res["sustained"] = {"steps": more, "seconds": dt, "value": wl.units_per_step * more / dt, "ms_per_step": dt / more * 1e3,
                            "kernel_ms_per_launch_overlapped": {k: v[0] / v[1] for k, v in fam_s.items() if v[1] > 0},
                            "note": "same workload, same lanes, stepped on for >= %.1f s; per-GPU figure (no collective inside)" % args.sustain_seconds}

    # ---- single stream: one lane, kernels back to back (their stand-alone durations) ----------
    if want_single and len(wl.contexts()) > 1:
        w1 = Workload(slam, torch, args, env.rank, env.local, 1, slots)
        for _ in range(args.warmup):
            step(w1)
        fence(w1, coll=False)
        e1, _, fam1, _ = timed(w1, args.steps, coll=False)
        res["single"] = {"lanes": 1, "ms_per_step": e1 / args.steps * 1e3, "value": w1.units_per_step * args.steps / e1,
                         "kernel_ms_per_launch": {k: v[0] / v[1] for k, v in fam1.items() if v[1] > 0}}
        # (start / stop events on every dispatch keep one stream's kernels from following each other back to back: the
        # same K steps without them)
        e0, _, _, _ = timed(w1, args.steps, coll=False, events=False)
        res["single"]["no_events"] = {"ms_per_step": e0 / args.steps * 1e3, "value": w1.units_per_step * args.steps / e0}
        w1.close()
        del w1
        if args.config != "particles" and not args.pipeline:
            # one trajectory at a time with the context's three-stream pipeline (scan matching | pose composition | map stage
            # of consecutive replays overlap): what ONE stream of replays reaches without a second context
            ap = argparse.Namespace(**vars(args))
            ap.pipeline = 1
            wp = Workload(slam, torch, ap, env.rank, env.local, 1, slots)
            for _ in range(args.warmup):
                step(wp)
            fence(wp, coll=False)
            ep, _, _, _ = timed(wp, args.steps, coll=False)
            res["single_pipelined"] = {"lanes": 1, "pipeline": 1, "ms_per_step": ep / args.steps * 1e3, "value": wp.units_per_step * args.steps / ep}
            ep0, _, _, _ = timed(wp, args.steps, coll=False, events=False)
            res["single_pipelined"]["no_events"] = {"ms_per_step": ep0 / args.steps * 1e3, "value": wp.units_per_step * args.steps / ep0}
            wp.close()
            del wp
    elif len(wl.contexts()) == 1:
        res["single"] = {"lanes": 1, "ms_per_step": elapsed_local / args.steps * 1e3, "value": wl.units_per_step * args.steps / elapsed_local,
                         "kernel_ms_per_launch": {k: v[0] / v[1] for k, v in fam.items() if v[1] > 0}, "note": "the main measurement is single-stream"}
        if want_single:
            e0, _, _, _ = timed(wl, args.steps, coll=False, events=False)
            res["single"]["no_events"] = {"ms_per_step": e0 / args.steps * 1e3, "value": wl.units_per_step * args.steps / e0}
    return res


def roofline_of(args, res):
    """The "roofline" object of the JSON line from a measurement's raw results (dominant kernel = largest share of the
    HIP-event time; its duration is the stand-alone one-lane duration wherever one was measured)."""
    wl, fam, single, elapsed = res["wl"], res["fam"], res["single"], res["elapsed"]
    ms = {k: v[0] for k, v in fam.items() if v[1] > 0}
    # the dominant family: by STAND-ALONE duration where the one-lane repeat measured it (every family is launched once per step;
    # overlapped durations also count the time a kernel waits for the chip beside the other lanes' kernels)
    alone = {k: v for k, v in ((single or {}).get("kernel_ms_per_launch") or {}).items() if k in ms}
    dom = max(alone, key=alone.get) if alone else max(ms, key=ms.get)
    dom_ms, dom_n = fam[dom]
    alg = wl.algorithmic_bytes()
    kname = wl.family_kernels.get(dom, dom)
    # the stand-alone duration of the dominant kernel where one exists: with several lanes a
    # kernel's duration includes the time it shares the chip with the other lanes' kernels
    avg_ms = dom_ms / dom_n
    alone_ms = single["kernel_ms_per_launch"].get(dom, avg_ms) if single else avg_ms
    pmc = load_pmc(args.config, kname)
    # (the PMC passes of tools/profile_gpu.sh run the configuration's defaults: scans - 1 pairs x trajectories per launch)
    profiled_units = (CONFIGS[args.config]["scans"] - 1) * CONFIGS[args.config].get("traj", 1) if args.config != "particles" else wl.units_per_step
    if pmc and wl.units_per_step != profiled_units:
        # the PMC passes ran the configuration's default size: counts per launch scale with the scans per launch
        scale = wl.units_per_step / float(profiled_units)
        for key in ("hbm_bytes_per_launch", "valu_insts_per_launch", "lds_insts_per_launch"):
            if pmc.get(key):
                pmc[key] = pmc[key] * scale
        pmc["source"] = "%s, scaled x%.3f to %d scans per launch" % (pmc.get("source"), scale, wl.units_per_step)
    hbm = {"achieved": alg.get(dom, 0) / (alone_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "algorithmic_bytes_per_launch": alg.get(dom, 0)}
    hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBS
    traffic = pmc.get("hbm_bytes_per_launch")
    if dom == "icp":
        # k_icp is bound by vector issue, not by HBM.  frac = the SIMD issue cycles of one
        # launch / (stand-alone launch duration x 1024 SIMDs x 2.4 GHz), the cycles from the hardware's own
        # instruction-class counters priced with measured costs (issue_cycles above); the figure that prices EVERY
        # instruction at the float64 cost of 4 cycles stays as all_f64_pricing.
        insts = pmc.get("valu_insts_per_launch")
        cyc, detail = issue_cycles(pmc)
        if cyc is None and insts and pmc.get("issue_mix"):
            # (PMC passes of an older tree, without the instruction-class counters: the static float64 share at 4 cycles, the rest at 2)
            share = pmc["issue_mix"]["f64_share"]
            cyc, detail = insts * (4.0 * share + 2.0 * (1.0 - share)), {"fallback": "static ISA mix (tools/isa_mix.py): float64 4 cycles, the rest 2; re-run tools/profile_gpu.sh for the hardware's class counters"}
        ach = cyc / (alone_ms * 1e-3) if cyc else None
        all64 = insts / (alone_ms * 1e-3) / F64_ISSUE_PEAK if insts else None
        roofline = {"kernel": kname, "bound": "valu_issue", "achieved": ach, "peak": SIMD_CYCLES_PER_S, "unit": "SIMD issue cycles/s",
                    "frac": ach / SIMD_CYCLES_PER_S if ach else None, "traffic": traffic,
                    "issue": detail, "all_f64_pricing": {"frac": all64, "note": "SQ_INSTS_VALU x 4 cycles: an upper bound of the issue time"},
                    "valu_busy_frac_pmc": pmc.get("valu_busy_frac"), "hbm": hbm,
                    "note": "achieved = issue cycles per launch (instruction-class counters of the committed PMC pass x measured cost per class) / stand-alone launch duration measured live"}
    elif kname.split("+")[0] in ("k_grid_update_win", "k_wedge_sort", "k_ray_bits"):
        # the LDS-window ray casts: every cell visit is one ds_add_u32 into the window; HBM sees the flushed windows only
        visits_s = wl.visits / (alone_ms * 1e-3)
        roofline = {"kernel": kname, "bound": "lds_atomic", "achieved": visits_s, "peak": LDS_ATOMIC_PEAK, "unit": "cell visits/s",
                    "frac": visits_s / LDS_ATOMIC_PEAK, "traffic": traffic,
                    "hbm_algorithmic": dict(hbm, note="9 B per cell visit / duration against the 8 TB/s spec: bytes the LDS window absorbs - HBM sees `traffic`"),
                    "lds_conflict_cycle_share": pmc.get("lds_conflict_cycle_share"),
                    "note": "achieved = in-bounds cell visits per launch / stand-alone duration of the whole launch (walk AND its set-up, sort, zero, flush phases); "
                            "peak = ds_add_u32 rate of 256 CUs for the walk's address pattern (tools/ubench_issue.hip: 5.95 per cycle per CU with 16 waves); "
                            "with two workgroups per CU the walk's vector issue is level with its LDS atomics (profiles/r05_dropped_experiments.txt #6, DESIGN.md K4a)"}
    else:
        roofline = dict(hbm, kernel=kname, bound="hbm", traffic=traffic)
        if traffic and alone_ms:
            # what the chip delivers for this kind of traffic (tools/ubench_rmw.hip, profiles/): counters read and written
            # back in place; the physical rate is the counter-measured bytes over the same duration
            roofline["physical"] = {"achieved": traffic / (alone_ms * 1e-3) / 1e9, "unit": "GB/s", "traffic_over_algorithmic": traffic / max(alg.get(dom, 0), 1),
                                    "mixed_rw_peak_measured": MIXED_RW_PEAK_GBS,
                                    "frac_of_mixed_rw_peak": traffic / (alone_ms * 1e-3) / 1e9 / MIXED_RW_PEAK_GBS}
    if dom != "icp" and fam.get("icp", (0.0, 0))[1]:
        # the scan matcher of a configuration another kernel dominates, priced exactly as the replay's roofline.frac: in the
        # particle batch its 10 000 pairs fill the chip, which the lone 999-pair launch of the replay does not
        pk = load_pmc(args.config, "k_icp")
        if pk.get("issue_mix_hw") and not pk.get("issue_mix"):
            mix = load_pmc("replay", "k_icp").get("issue_mix")       # (the static classes of the same kernel: tools/isa_mix.py)
            if mix:
                q3 = (mix.get("qpt3") or {}).get("classes")
                pk = dict(pk, issue_mix=dict(mix, static_classes=q3) if q3 else mix)
        cyc_i, det_i = issue_cycles(pk)
        icp_ms = (single["kernel_ms_per_launch"].get("icp") if single else None) or fam["icp"][0] / fam["icp"][1]
        if cyc_i and icp_ms:
            roofline["icp_issue"] = {"kernel": "k_icp", "bound": "valu_issue", "frac": cyc_i / (icp_ms * 1e-3) / SIMD_CYCLES_PER_S, "avg_launch_ms": icp_ms,
                                     "instructions": pk.get("valu_insts_per_launch"), "cycles_per_instruction": det_i["cycles_per_instruction"],
                                     "lanes": 1 if single else len(wl.contexts()),
                                     "note": "SIMD issue cycles per launch (this configuration's instruction-class counters x measured costs; static classes of the three-queries shape) / stand-alone launch duration x 1024 SIMDs x 2.4 GHz"}
    if pmc.get("stale"):
        roofline["stale_pmc"] = pmc["stale"]
    roofline.update({"traffic_source": pmc.get("source"), "avg_launch_ms": alone_ms, "avg_launch_ms_overlapped": avg_ms, "launches": dom_n,
                     "kernel_ms_per_launch_overlapped": {k: fam[k][0] / fam[k][1] for k in ms},
                     "timed": "start/stop HIP events carried by the dispatches of all %d lanes; avg_launch_ms is the one-lane (stand-alone) duration, "
                              "the overlapped durations include the time a kernel shares the chip with the other lanes' kernels" % len(wl.contexts())})
    if res.get("fam_note"):
        roofline["timed_families"] = res["fam_note"]
    roofline["lanes"] = 1 if single else len(wl.contexts())          # lanes the duration behind `frac` was measured at
    if hasattr(wl, "family_kernels"):
        # What the CHIP issued over the timed region, all kernels of all lanes together: issue cycles per launch of every
        # family (its instruction-class counters priced as above; k_icp in the launch shape this run used) x launches /
        # (duration of the timed region x 1024 SIMDs x 2.4 GHz) - the one figure that describes the run `value` comes from
        tot, tot_all64, missing = 0.0, 0.0, []
        for f, (t_ms, n_l) in fam.items():
            if not n_l:
                continue
            kk = wl.family_kernels.get(f)
            pk = load_pmc(args.config, kk) if kk else {}
            per = pk.get("valu_insts_per_launch_qpt3") if f == "icp" and len(wl.contexts()) > 1 and pk.get("valu_insts_per_launch_qpt3") else pk.get("valu_insts_per_launch")
            if pmc and wl.units_per_step != profiled_units and per:
                per = per * wl.units_per_step / float(profiled_units)
            cyc_f, _ = issue_cycles(pk, per) if per else (None, None)
            if per is None or cyc_f is None:
                missing.append(f)
            else:
                tot += cyc_f * n_l
                tot_all64 += per * 4.0 * n_l
        if not missing and elapsed > 0:
            roofline["chip"] = {"frac": tot / elapsed / SIMD_CYCLES_PER_S, "lanes": len(wl.contexts()), "unit": "share of the chip's SIMD issue cycles over the timed region",
                                "all_f64_pricing_frac": tot_all64 / elapsed / SIMD_CYCLES_PER_S,
                                "note": "sum over ALL kernels of issue cycles per launch (instruction-class counters, profiles/) x launches in the timed region / its duration"}
        elif missing:
            roofline["chip"] = {"frac": None, "missing_counters": missing}
    iters = np.asarray(wl.iters)
    if "icp" in fam and fam["icp"][1]:
        # The figure below counts the distance evaluations an EXHAUSTIVE nearest-neighbour scan would
        # make (iters * n_src * n_tar, what the reference does); the pruned search returns the same
        # result evaluating a fraction of them.
        evals = float(iters.astype(np.int64).sum()) * args.beams * args.beams
        icp_s = (single["kernel_ms_per_launch"].get("icp") if single else None) or fam["icp"][0] / fam["icp"][1]
        icp_s *= 1e-3
        roofline["icp_work"] = {"exhaustive_equivalent_distance_evals_per_s": evals / icp_s, "mean_iters": float(iters.mean()),
                                "note": "equivalent brute-force rate; f64 VALU peak is %.1f TFLOP/s (~%.1e evals/s at 6 flop each)"
                                        % (F64_VALU_PEAK_TFLOPS, F64_VALU_PEAK_TFLOPS * 1e12 / 6)}
        lds_insts = load_pmc(args.config, "k_icp").get("lds_insts_per_launch")
        if lds_insts:
            roofline["icp_work"]["lds"] = {"wave_instructions_per_launch": lds_insts, "upper_bound_GBps": lds_insts * 1024.0 / icp_s / 1e9,
                                           "peak_GBps": 256 * 128 * 2.4,
                                           "note": "most are per-lane ds_read_b128 of target points; box reads are broadcasts"}
    if "grid" in fam and fam["grid"][1]:
        grid_s = ((single["kernel_ms_per_launch"].get("grid") if single else None) or fam["grid"][0] / fam["grid"][1]) * 1e-3
        roofline["grid_cast"] = {"cell_visits_per_step": wl.visits, "visits_per_s": wl.visits / grid_s,
                                 "algorithmic_GBps": 9 * wl.visits / grid_s / 1e9, "frac_of_hbm_peak": 9 * wl.visits / grid_s / 1e9 / HBM_PEAK_GBS}
    return roofline


def assemble_line(args, n_ranks, value, ms_per_step, enqueue_ms, closing_ms, workload, units_per_step, lanes, roofline, use_dist,
                  single=None, sustained=None, parity=None, cpu_baseline=None, other_configs=None, single_gpu_same_workload=None,
                  rccl_world_size=None, timing_mask=None, instrumented=None, single_pipelined=None):
    """The ONE JSON line (a dict) from measured numbers.  Pure: tests/test_bench_cpu.py checks the schema of the N = 1
    and N > 1 lines with made-up measurements."""
    out = {
        "metric": "scans/sec (360-beam ICP + 0.05 m grid update)" if args.config != "dense" else "scans/sec (1080-beam ICP + 0.02 m grid update)",
        "value": value, "unit": "scans/s",
        "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "host_enqueue_ms_per_step": enqueue_ms,
        "closing_collectives_ms": closing_ms,   # all_gather + barrier after the last replay
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "name": args.config,
                   "units_per_step_per_gpu": units_per_step, "point_buffers": args.points,
                   "point_buffers_note": "storage type of the points the scan matcher sees (arithmetic is float64 either way)",
                   "pipeline": args.pipeline, "lanes": lanes, "trajectories_per_step": getattr(args, "traj", 1), "grid_mode": args.grid_mode, "grid_group": args.grid_group,
                   "parallelism": "1 trajectory per GPU" + (" (%d independent replays of it per step)" % getattr(args, "traj", 1) if getattr(args, "traj", 1) > 1 else "") +
                                  (", all_gather of final poses (%s)" % args.gather if use_dist else "")},
        "roofline": roofline,
    }
    if timing_mask is not None:
        out["timing_mask"] = timing_mask            # kernel families whose dispatches carried HIP events in the timed K steps
    if instrumented:
        out["instrumented"] = instrumented
    if single:
        out["single_stream"] = single
    if single_pipelined:
        out["single_stream_pipelined"] = single_pipelined
    if sustained:
        out["sustained"] = sustained
    if parity is not None:
        out["parity"] = parity
    if cpu_baseline is not None:
        out["cpu_baseline"] = cpu_baseline
    if other_configs:
        out["other_configs"] = other_configs
    if use_dist:
        # N > 1 runs configs[3] (5 000-scan trajectories), N = 1 runs configs[1] (1 000 scans): the per-GPU rates differ by
        # construction, so a scaling efficiency is value / (n_gpus x single_gpu_same_workload.value), not value / (n x the N = 1 line)
        out["rccl_world_size"] = rccl_world_size
        out["single_gpu_same_workload"] = single_gpu_same_workload
        if single_gpu_same_workload and single_gpu_same_workload.get("value"):
            out["scaling_efficiency_same_workload"] = value / (n_ranks * single_gpu_same_workload["value"])
    return out


def other_config_summary(args, res, roofline, parity):
    """What the default line carries of another configuration (bench.py --config <name> prints its full line)."""
    r = {"value": res["wl"].units_per_step * args.steps / res["elapsed"], "unit": "scans/s", "ms_per_step": res["elapsed"] / args.steps * 1e3,
         "steps": args.steps, "warmup": args.warmup, "lanes": len(res["wl"].contexts()), "workload": res["wl"].workload_name(),
         "roofline": {k: roofline.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "lanes", "physical", "hbm_algorithmic", "lds_conflict_cycle_share", "chip", "icp_issue", "stale_pmc") if k in roofline},
         "parity": parity}
    if res["single"]:
        r["single_stream"] = res["single"]
    if res.get("single_pipelined"):
        r["single_stream_pipelined"] = res["single_pipelined"]
    if res["sustained"]:
        r["sustained"] = {k: res["sustained"][k] for k in ("value", "ms_per_step", "steps", "seconds")}
    r["timing_mask"] = res.get("timing_mask")
    if res.get("instrumented"):
        r["instrumented"] = res["instrumented"]
    return r


def _child_leg(extra, roofline=True):
    """A secondary measurement in a CHILD process: `python bench.py <extra> --no-cpu-baseline --no-other-configs`, its JSON line
    condensed like other_config_summary.  For the 4-lane legs: they run a fifth slower at the end of a process that has replayed
    32 trajectories per call on two other contexts first (dense 1.47-1.56 against 1.65 M scans/s in a process of its own, one
    trajectory on 4 contexts 9.0 against 11.4 M; two-lane legs are not affected) - the figure a user of `bench.py --config dense`
    gets is the one of a fresh process.  This process only waits meanwhile (it starts no other program in place of itself)."""
    cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--no-cpu-baseline", "--no-other-configs"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode or not lines:
        return {"error": "child rc %d: %s" % (p.returncode, p.stderr[-300:])}
    d = json.loads(lines[-1])
    r = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "lanes": d["config"]["lanes"],
         "trajectories_per_step": d["config"].get("trajectories_per_step"), "workload": d["config"]["workload"],
         "roofline": {k: d["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "lanes", "physical", "hbm_algorithmic",
                                                        "lds_conflict_cycle_share", "chip", "icp_issue", "stale_pmc", "kernel_ms_per_launch_overlapped") if k in d["roofline"]},
         "parity": d.get("parity"), "process": "child (python bench.py %s)" % " ".join(extra)}
    if not roofline:            # (a leg without the one-lane repeat has no stand-alone durations to price a roofline with)
        r["kernel_ms_per_launch_overlapped"] = r.pop("roofline").get("kernel_ms_per_launch_overlapped")
    for k_out, k_in in (("single_stream", "single_stream"), ("single_stream_pipelined", "single_stream_pipelined"), ("timing_mask", "timing_mask"), ("instrumented", "instrumented")):
        if d.get(k_in) is not None:
            r[k_out] = d[k_in]
    if d.get("sustained"):
        r["sustained"] = {k: d["sustained"][k] for k in ("value", "ms_per_step", "steps", "seconds")}
    return r


def _replay_legs(args, env, torch, others):
    """configs[1] again, (a) as rounds 1-4 measured it: ONE trajectory per call on four overlapping contexts, (b) at the W7 launch
    file's scan-matching parameters (SURVEY.md 8d asks for both sets: W7_Dead Reckoning (ICP)/course_agv_slam/launch/icp.launch:10-12
    sets max_iter 10, tolerance 0 - every pair runs exactly ten iterations)."""
    base = ["--max-iter", str(args.max_iter), "--tol", repr(args.tol), "--sustain-seconds", "0.5"] + (["--no-parity"] if args.no_parity else [])
    others["lanes4_single_trajectory"] = _child_leg(["--config", "replay", "--traj", "1", "--lanes", "4", "--grid-group", "0", "--steps", "48", "--warmup", "5",
                                                     "--no-single-stream"] + base, roofline=False)
    for name, change in (("replay_w7_params", dict(max_iter=10, tol=0.0, steps=12 - 12 % max(args.lanes, 1) or args.lanes)),):
        a4 = argparse.Namespace(**vars(args))
        for k, v in change.items():
            setattr(a4, k, v)
        a4.sustain_seconds = 0.5
        try:
            r4 = measure(a4, env, want_single=(name == "replay_w7_params"), want_sustained=True)
            w4 = r4["wl"]
            o4 = {"value": w4.units_per_step * a4.steps / r4["elapsed"], "unit": "scans/s", "ms_per_step": r4["elapsed"] / a4.steps * 1e3, "steps": a4.steps,
                  "warmup": a4.warmup, "lanes": len(w4.contexts()), "trajectories_per_step": a4.traj, "workload": w4.workload_name(),
                  "sustained": {k: r4["sustained"][k] for k in ("value", "ms_per_step", "steps", "seconds")} if r4["sustained"] else None,
                  "kernel_ms_per_launch_overlapped": {k: v[0] / v[1] for k, v in r4["fam"].items() if v[1] > 0},
                  "mean_iters": float(np.asarray(r4["dev_results"][0]["iters"]).mean()),
                  "parity": None if args.no_parity else w4.parity(r4["dev_results"])}
            if r4["single"]:
                o4["single_stream"] = r4["single"]
            others[name] = o4
            w4.close()
            del r4, w4
        except Exception as e:
            others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or unset WORLD_SIZE and "
              "let bench.py start the ranks)" % (args.gpus, world, args.gpus), file=sys.stderr)
        sys.exit(2)
    # Each lane's stream should own a hardware queue (the HIP runtime multiplexes streams onto
    # GPU_MAX_HW_QUEUES queues, 4 by default): with a process group RCCL adds streams of its own,
    # and two lanes sharing a queue serialise (measured 0.26 instead of 0.18 ms per step).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import importlib
    import torch
    dist = None
    # under torch.distributed.run (RANK and MASTER_ADDR set) the collective path is taken even
    # for one rank, so that it can be rehearsed on a one-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    slam = importlib.import_module(PKG)
    same_workload = None
    if use_dist:
        torch.cuda.set_device(local)
        # BEFORE the process group exists: rank 0 times its own share of the N-GPU workload alone (same scans, same lanes,
        # no collective), so that the line carries the one-GPU figure of the SAME workload
        if rank == 0 and not args.no_timing:
            env1 = Env(slam, torch, None, 0, 1, local, False)
            r1 = measure(args, env1, collective=False, want_single=True, want_sustained=False)
            same_workload = {"value": r1["wl"].units_per_step * args.steps / r1["elapsed"], "unit": "scans/s", "ms_per_step": r1["elapsed"] / args.steps * 1e3,
                             "workload": r1["wl"].workload_name(), "single_stream": r1["single"],
                             "note": "rank 0 alone, before init_process_group: the same 5 000-scan trajectory, lanes and steps"}
            same_single = r1["single"]
            r1["wl"].close()
            del r1
            torch.cuda.empty_cache()
        import torch.distributed as dist
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout
        # for the ONE JSON line (the banner goes to stderr instead)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    for li in range(4):                 # the lanes' streams first, before any other stream of the process (see lane_stream)
        lane_stream(torch, local, li, 4)
    env = Env(slam, torch, dist, rank, world, local, use_dist)
    cpu = cpu_baseline_first(slam, args, rank) if rank == 0 and world == 1 and not args.no_cpu_baseline else None
    res = measure(args, env, want_single=not args.no_single_stream and not use_dist, want_sustained=not use_dist)
    wl = res["wl"]
    n_ranks = dist.get_world_size() if use_dist else 1
    value = wl.units_per_step * n_ranks * args.steps / res["elapsed"]

    if args.no_timing:
        if rank == 0:
            print(json.dumps({"value": value, "ms_per_step": res["elapsed"] / args.steps * 1e3, "lanes": args.lanes, "traj": args.traj,
                              "pipeline": args.pipeline, "note": "experiment without HIP-event timing"}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    if use_dist and same_workload and rank == 0:
        res["single"] = same_single          # stand-alone kernel durations of the same workload: roofline.frac keeps its meaning under a process group
    roofline = roofline_of(args, res)
    parity = wl.parity(res["dev_results"]) if rank == 0 and not args.no_parity else None
    single, sustained = res["single"], res["sustained"]

    # ---- the other single-GPU configurations of BASELINE.json, briefly, in the same process (outside every timed region
    #      above): configs[2] (particles) and configs[4] (dense), each with its own roofline and parity block ----------
    others = None
    if args.config == "replay" and not use_dist and not args.no_other_configs and rank == 0:
        res["workload_name"], res["units"], res["lanes"] = wl.workload_name(), wl.units_per_step, len(wl.contexts())
        wl.close()
        del res["wl"], wl
        torch.cuda.empty_cache()
        others = {}
        _replay_legs(args, env, torch, others)
        for name in ("particles", "dense"):
            a2 = config_args(args, name)
            if name == "dense":                     # four lanes: in a process of its own (_child_leg)
                others[name] = _child_leg(["--config", "dense", "--max-iter", str(args.max_iter), "--tol", repr(args.tol), "--sustain-seconds", "0.5"] +
                                          (["--no-parity"] if args.no_parity else []))
                continue
            try:
                r2 = measure(a2, env, want_single=True, want_sustained=True)
                others[name] = other_config_summary(a2, r2, roofline_of(a2, r2), None if args.no_parity else r2["wl"].parity(r2["dev_results"]))
                r2["wl"].close()
                del r2
                if name == "particles":
                    # the same batch with the hypotheses' previous poses scattered (the reference has no particle filter: SURVEY.md
                    # 8d cfg3 starts every hypothesis at one pose; a filter's particles differ): boxes of all shapes, rays that
                    # leave a map, the general owner kernel behind the byte-window one
                    torch.cuda.empty_cache()
                    a3 = argparse.Namespace(**vars(a2))
                    a3.pose_spread = 0.5
                    r3 = measure(a3, env, want_single=True, want_sustained=False)
                    others[name]["scattered_poses"] = {"pose_spread": 0.5, "value": r3["wl"].units_per_step * a3.steps / r3["elapsed"], "unit": "scans/s",
                                                       "ms_per_step": r3["elapsed"] / a3.steps * 1e3, "single_stream": r3["single"],
                                                       "parity": None if args.no_parity else r3["wl"].parity(r3["dev_results"]),
                                                       "note": "previous poses drawn from N(0, 0.5) in x, y and heading (as tests/test_gpu_configs.py::test_config2_10000_particles does)"}
                    r3["wl"].close()
                    del r3
            except Exception as e:                      # the headline line must not die of a secondary configuration
                others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        wl_name, units, lanes = res["workload_name"], res["units"], res["lanes"]
    if rank == 0:
        if others is None:
            wl_name, units, lanes = wl.workload_name(), wl.units_per_step, len(wl.contexts())
        out = assemble_line(args, n_ranks, value, res["elapsed"] / args.steps * 1e3, res["enqueue"] / args.steps * 1e3, res["closing_ms"],
                            wl_name, units, lanes, roofline, use_dist, single=single, sustained=sustained, parity=parity, cpu_baseline=cpu,
                            other_configs=others, single_gpu_same_workload=same_workload,
                            rccl_world_size=n_ranks if use_dist else None, timing_mask=res.get("timing_mask"), instrumented=res.get("instrumented"),
                            single_pipelined=res.get("single_pipelined"))
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
