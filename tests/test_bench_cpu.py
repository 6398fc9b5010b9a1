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


def _fake_args(**kw):
    import argparse
    d = dict(config="replay", steps=20, warmup=5, points="f64", pipeline=0, grid_mode=1, grid_group=0, gather="end", gpus=1)
    d.update(kw)
    return argparse.Namespace(**d)


def test_json_line_schema_single_and_multi_gpu():
    """The JSON line is assembled by a pure function: with made-up measurements the N = 1 line carries the contract's
    keys plus roofline / cpu_baseline / other_configs, and the N > 1 line carries what makes it comparable with one
    GPU: the one-GPU figure of the SAME 5 000-scan workload, the RCCL world size, and stand-alone
    kernel durations behind roofline.frac."""
    sys.path.insert(0, ROOT)
    import bench
    roof = {"kernel": "k_icp", "bound": "valu_f64_issue", "achieved": 2.8e11, "peak": 6.1e11, "unit": "wave-instructions/s", "frac": 0.46, "traffic": 3.5e6}
    one = bench.assemble_line(_fake_args(), 1, 8.5e6, 0.118, 0.05, 0.2, "configs[1]: ...", 999, 4, roof, False,
                              single={"lanes": 1, "ms_per_step": 0.26, "value": 3.8e6, "kernel_ms_per_launch": {"icp": 0.12}},
                              sustained={"steps": 4000, "seconds": 0.5, "value": 9.8e6, "ms_per_step": 0.1, "kernel_ms_per_launch_overlapped": {"icp": 0.23}},
                              parity={"iters_equal": True}, cpu_baseline={"value": 3000.0, "unit": "scans/s", "cores": 64, "kind": "port", "sample": "..."},
                              other_configs={"particles": {"value": 7.6e6, "roofline": {"frac": 0.34, "traffic": 3.1e9}, "parity": {}},
                                             "dense": {"value": 8.5e5, "roofline": {"frac": 0.38, "traffic": 3.8e9}, "parity": {}}})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity", "single_stream", "sustained", "other_configs"):
        assert k in one, k
    assert one["n_gpus"] == 1 and one["vs_baseline"] is None and one["config"]["workload"].startswith("configs[1]")
    assert set(one["other_configs"]) == {"particles", "dense"} and "kernel_ms_per_launch_overlapped" in one["sustained"]
    assert "single_gpu_same_workload" not in one and "rccl_world_size" not in one
    same = {"value": 9.0e6, "unit": "scans/s", "ms_per_step": 0.55, "workload": "configs[3] share: ...",
            "single_stream": {"lanes": 1, "kernel_ms_per_launch": {"icp": 0.6}}}
    many = bench.assemble_line(_fake_args(gpus=8, steps=12, warmup=2), 8, 6.8e7, 0.59, 0.1, 0.3, "configs[3] share: ...", 4999, 4, roof, True,
                               single=same["single_stream"], single_gpu_same_workload=same, rccl_world_size=8)
    for k in ("single_gpu_same_workload", "rccl_world_size", "scaling_efficiency_same_workload", "single_stream", "roofline"):
        assert k in many, k
    assert many["n_gpus"] == 8 and many["rccl_world_size"] == 8 and many["config"]["workload"].startswith("configs[3]")
    assert abs(many["scaling_efficiency_same_workload"] - 6.8e7 / (8 * 9.0e6)) < 1e-12
    assert "all_gather" in many["config"]["parallelism"] and many["single_stream"]["kernel_ms_per_launch"]["icp"] == 0.6


def test_issue_cycles_prices_counted_instruction_classes():
    """bench.issue_cycles: a launch's SIMD issue cycles from the hardware's instruction-class counters and the measured cost
    per class; classes without a counter (float64 compares / min / max, DPP and lane exchanges) enter with their static
    share relative to the counted float64 arithmetic; everything else at the plain 32-bit cost."""
    sys.path.insert(0, ROOT)
    import bench
    pmc = {"valu_insts_per_launch": 1000.0,
           "issue_mix_hw": {"f64_arith": 300.0, "f64_trans": 10.0, "cvt": 20.0, "trans_f32": 5.0, "int32": 100.0, "source": "x"},
           "issue_mix": {"static_classes": {"f64_arith": 3000, "f64_cmp_minmax": 1500, "dpp_lane": 400, "cndmask": 200}}}
    cyc, detail = bench.issue_cycles(pmc)
    c = bench.ISSUE_COST
    n = detail["instructions"]
    assert n["f64_cmp_minmax_est"] == 150.0 and n["dpp_lane_est"] == 60.0 and abs(n["simple"] - (1000 - 300 - 10 - 20 - 5 - 150 - 60)) < 1e-9
    want = (300 + 150) * c["f64"] + (20 + 60) * c["quarter"] + 10 * c["trans_f64"] + 5 * c["trans_f32"] + n["simple"] * c["simple"]
    assert abs(cyc - want) < 1e-9 and abs(detail["cycles_per_instruction"] - want / 1000.0) < 1e-12
    # another launch shape of the same kernel (its own instruction count): the same mix, scaled
    cyc2, _ = bench.issue_cycles(pmc, 2000.0)
    assert abs(cyc2 - 2.0 * want) < 1e-6
    # without the class counters there is nothing to price
    assert bench.issue_cycles({"valu_insts_per_launch": 1000.0}) == (None, None)
    # a kernel without static classes: counted classes only
    cyc3, d3 = bench.issue_cycles({"valu_insts_per_launch": 100.0, "issue_mix_hw": {"f64_arith": 10.0, "f64_trans": 0.0, "cvt": 0.0, "trans_f32": 0.0}})
    assert abs(cyc3 - (10 * c["f64"] + 90 * c["simple"])) < 1e-9 and d3["instructions"]["f64_cmp_minmax_est"] == 0.0


def test_source_stamp_ignores_comments_and_blank_lines(tmp_path, monkeypatch):
    """The stamp that ties profiles/pmc_traffic.json to the kernel sources hashes code only: a comment or blank-line edit must
    not mark the committed PMC figures stale, a code edit must."""
    sys.path.insert(0, ROOT)
    import bench
    csrc = tmp_path / bench.PKG / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "a.hip").write_text("// header\nint f(int x)\n{\n    return x + 1;   // add one\n}\n")
    (csrc / "b.h").write_text("/* block\n   comment */\n#pragma once\nconstexpr int k = 3;\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    h0 = bench.source_hash()
    (csrc / "a.hip").write_text("// another header, longer\n\n\nint f(int x)\n{\n    return x + 1;       /* add one */\n}\n\n")
    assert bench.source_hash() == h0
    (csrc / "a.hip").write_text("int f(int x)\n{\n    return x + 2;\n}\n")
    assert bench.source_hash() != h0
