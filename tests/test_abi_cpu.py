"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol that
include/slam_hip.h declares, and refuses to run without a GPU (no CPU fallback); plus the
host logic that needs no device (parameters, synthetic streams, index-rule helpers)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import PKG, ROOT, pkg


def test_library_exports_every_declared_symbol():
    abi = pkg("_abi")
    L = abi.lib()
    names = abi.header_symbols()
    assert len(names) >= 30 and "slam_replay_dev" in names and "slam_icp_batch" in names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(abi._SIGS) == set(names)          # the ctypes binding covers the whole header
    assert L.slam_abi_version() == 1


def test_header_cites_reference_for_every_operator():
    text = open(os.path.join(ROOT, "include", "slam_hip.h")).read()
    for ref in ("icp.py:38-88", "icp.py:90-114", "icp.py:149-179", "mapping.py:22-51", "mapping.py:8-20",
                "bresenham.py:2-58", "slam_ekf.py:63-95", "slam_ekf.py:270-271", "icp.py:153-158"):
        assert ref in text, ref
    assert not re.search(r"\btorch\b|at::Tensor", text.split("Conventions")[1].split("#ifndef")[0].replace("torch.Tensor.data_ptr()", ""))


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    slam = pkg()
    with pytest.raises(slam.SlamError, match="no HIP device|no CPU path"):
        slam.ICP()
    with pytest.raises(slam.SlamError):
        slam.Mapping(200, 200, 0.1)
    with pytest.raises(slam.SlamError):
        slam.bresenham([0, 0], [3, 1])
    with pytest.raises(slam.SlamError):
        slam.DeviceReplay(np.zeros((3, 8), dtype=np.float32), -3.0, 3.0)
    # identical endpoints never reach the device (bresenham.py:10-11)
    assert slam.bresenham([2, 2], [2, 2]).path == []


def test_product_never_uses_the_oracle():
    """Nothing under the package imports, loads or links anything from oracle/."""
    pkg_dir = os.path.join(ROOT, PKG)
    pat = re.compile(r"import\s+oracle|from\s+oracle|liboracle|oracle[/\\]|c_oracle|oracle_np|slam_oracle")
    for dp, _dn, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert not pat.search(txt), f


def test_null_and_bad_arguments_are_rejected_before_any_device_work():
    abi = pkg("_abi")
    L = abi.lib()
    assert L.slam_create(0, None, None) == abi.ERR_INVALID
    assert b"out is null" in L.slam_last_error()
    assert L.slam_synchronize(None) == abi.ERR_INVALID
    assert L.slam_destroy(None) == abi.SLAM_OK
    h = C.c_void_p()
    assert L.slam_grid_create(None, 1, 10, 10, 10.0, 10.0, 10.0, 0.01, 20.0, 10.0, C.byref(h)) == abi.ERR_INVALID


def test_params_table():
    p = pkg("param")
    p.clear_params()
    assert p.get_param('/icp/max_iter', 30) == 30
    p.set_param('/icp/max_iter', 10)
    assert p.get_param('/icp/max_iter', 30) == 10
    p.clear_params()
    with pytest.raises(KeyError):
        p.get_param('/slam/map_width')


def test_synthetic_streams_are_deterministic_and_in_spec(syn):
    a, b = syn.make_replay(12, 120, seed=5, stride=5), syn.make_replay(12, 120, seed=5, stride=5)
    assert np.array_equal(a.ranges, b.ranges) and a.ranges.dtype == np.float32 and a.ranges.shape == (12, 120)
    assert a.ranges.min() >= 0.1 and a.ranges.max() <= 30.0
    assert not np.array_equal(a.ranges, syn.make_replay(12, 120, seed=6, stride=5).ranges)
    step = np.hypot(*np.diff(a.poses_true[:, :2], axis=0).T)
    assert np.allclose(step, 0.15, atol=2e-3)            # 0.3 m/s, every 5th message of 10 Hz
    m = a.message(3)
    assert isinstance(m.ranges, tuple) and m.angle_min == -3.14159 and len(m.ranges) == 120
    pri = syn.particle_priors(1000, seed=2)
    assert abs(pri[:, 0].std() - 0.05) < 0.005 and abs(pri[:, 2].std() - np.deg2rad(2)) < 0.004


def test_trig_tables_match_reference_formula():
    abi = pkg("_abi")
    ct, st = abi.trig_tables(-3.14159, 3.14159, 360)
    ang = np.linspace(-3.14159, 3.14159, 360)
    assert np.array_equal(ct, np.cos(ang)) and np.array_equal(st, np.sin(ang))
