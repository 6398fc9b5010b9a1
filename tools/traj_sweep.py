#!/usr/bin/env python3
"""Sweep of configs[1] as batches of L trajectories per slam_replay_dev call x contexts (lanes): which packing fills the
chip (VERDICT r4 "next" #1).  Runs bench.py once per point as a child process (nothing here touches the GPU) with the
secondary legs off, and prints one line per point: value of the K steps, the sustained figure, the overlapped kernel
durations.  `python tools/traj_sweep.py [--config replay] [--points "1x4,4x1,4x2,8x1,8x2"] [--extra "..."]`."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="replay")
    ap.add_argument("--points", default="1x1,1x4,2x1,2x2,4x1,4x2,8x1,8x2,16x1")
    ap.add_argument("--steps", type=int, default=0, help="0: 48 / L (at least 8) per point")
    ap.add_argument("--extra", default="", help="further bench.py arguments for every point")
    ap.add_argument("--single", action="store_true", help="keep the one-lane repeat (stand-alone kernel durations)")
    a = ap.parse_args()
    for pt in a.points.split(","):
        L, lanes = (int(v) for v in pt.split("x"))
        steps = a.steps or max(8, (48 // L) - (48 // L) % lanes)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", a.config, "--traj", str(L), "--lanes", str(lanes), "--steps", str(steps),
               "--warmup", str(max(2, lanes)), "--no-cpu-baseline", "--no-other-configs", "--no-parity", "--sustain-seconds", "0.5"]
        if not a.single:
            cmd.append("--no-single-stream")
        cmd += a.extra.split()
        p = subprocess.run(cmd, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode or not line:
            print("L=%d lanes=%d: FAILED rc=%d %s" % (L, lanes, p.returncode, p.stderr[-400:]), flush=True)
            continue
        d = json.loads(line[-1])
        su = d.get("sustained") or {}
        ins = (d.get("instrumented") or {}).get("no_events") or {}
        row = {"L": L, "lanes": lanes, "steps": steps, "value_M": round(d["value"] / 1e6, 3), "ms_per_step": round(d["ms_per_step"], 4),
               "sustained_M": round(su.get("value", 0) / 1e6, 3), "no_events_M": round(ins.get("value", 0) / 1e6, 3),
               "kernel_ms_overlapped": {k: round(v, 4) for k, v in (su.get("kernel_ms_per_launch_overlapped") or d["roofline"].get("kernel_ms_per_launch_overlapped") or {}).items()}}
        if a.single and d.get("single_stream"):
            row["single_kernel_ms"] = {k: round(v, 4) for k, v in d["single_stream"].get("kernel_ms_per_launch", {}).items()}
            row["single_M"] = round(d["single_stream"]["value"] / 1e6, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
