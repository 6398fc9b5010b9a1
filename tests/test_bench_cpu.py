"""bench.py's multi-GPU entry point, checked without a GPU: `--gpus N` outside
torch.distributed.run must start the ranks as CHILD processes from a parent that has not
imported torch (a process that has initialised HIP must never be replaced or re-used as a
launcher), and a mismatch between --gpus and WORLD_SIZE must fail loudly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_ranks_from_a_torch_free_parent(tmp_path):
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import sys, json, subprocess\n"
        "sys.argv = ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1']\n"
        "sys.path.insert(0, %r)\n"
        "calls = []\n"
        "def fake_call(cmd, env=None):\n"
        "    calls.append((cmd, 'torch' in sys.modules, env.get('HSA_ENABLE_IPC_MODE_LEGACY')))\n"
        "    return 7\n"
        "subprocess.call = fake_call\n"
        "import bench\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    print(json.dumps({'rc': e.code, 'calls': calls, 'torch_loaded': 'torch' in sys.modules}))\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(probe)], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    import json
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["rc"] == 7 and not d["torch_loaded"]                  # the children's status is relayed
    (cmd, torch_loaded, ipc), = d["calls"]
    assert not torch_loaded and ipc == "0"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")


def test_gpus_world_size_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                         env=env, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr and out.stdout.strip() == ""
