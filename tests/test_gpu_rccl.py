"""The RCCL path of the multi-GPU configuration (BASELINE.json configs[3]: one trajectory per rank, ONE all_gather of
final poses; the trajectories are independent - W12m/slam_ekf.py:109-113 - so nothing else is exchanged), run for
real on the one GPU of the test box: bench.py under `torch.distributed.run --nproc-per-node 1` takes the collective
path (init_process_group("nccl"), all_gather_into_tensor, barrier, all_reduce(MAX) of the elapsed time) with a world
of one rank.  The launcher and the rank are CHILD processes of this test; the launcher starts the rank before
anything in it has touched the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_under_torch_distributed_run_takes_the_rccl_path():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "4",
           "--no-cpu-baseline", "--no-other-configs"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                  # ONE JSON line; RCCL's banner went to stderr
    d = json.loads(lines[0])
    assert d["rccl_world_size"] == 1 and d["n_gpus"] == 1
    assert d["config"]["parallelism"].endswith("all_gather of final poses (end)")
    assert d["single_gpu_same_workload"]["value"] > 0 and d["scaling_efficiency_same_workload"] > 0
    assert d["closing_collectives_ms"] is not None and d["closing_collectives_ms"] >= 0.0
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    par = d["parity"]
    assert par["iters_equal"] and par["counter_cell_mismatches"] == 0 and par["pmap_cell_mismatches"] == 0 and par["visits_equal"]
    assert par["pose_max_abs_err"] < 1e-9 and par["T_max_abs_err"] < 1e-9


def _run_child(script, timeout=900):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_shared_map_merge_is_stream_ordered_under_rccl():
    """SURVEY.md 8e, shared-map case (additive evidence, W12m/mapping.py:42-50): dist.all_reduce_grid inside a real RCCL
    process group (world 1: the one GPU of the box), right behind an ASYNCHRONOUS DeviceReplay.run() on a context whose
    stream is not the collectives' - tests/rccl_merge_child.py.  Until round 5 the merge ordered nothing against the
    context's stream and had never run inside a process group."""
    d = _run_child("rccl_merge_child.py")
    assert d["world"] == 1 and d["backend"] == "nccl"
    assert d["cells_touched"] > 10000
    assert d["merged_counter_mismatches"] == 0          # map 0 + map 1 == the map of the whole trajectory, read behind the merge
    assert d["reset_came_last"]                         # the context waited for torch's stream before its next update
    # (d["unordered_read_mismatches"] is informational: > 0 shows what the merge read before it was ordered)
