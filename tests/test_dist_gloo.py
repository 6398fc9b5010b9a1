"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercise the sharding,
the all_gather of final poses and the counter all_reduce.  The compute is the oracle (the
product's runner needs a GPU); what is under test is the multi-rank plumbing."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT, pkg

WORKER = r'''
import importlib, os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
slam_dist = importlib.import_module(%(pkg)r + ".dist")
syn = importlib.import_module(%(pkg)r + ".synthetic")
from oracle import c_oracle as co

def runner(ranges, amin, amax, max_iter, tol, local):
    return np.stack([co.replay(r, amin, amax, None, max_iter, tol)[0] for r in ranges])

n_traj = int(sys.argv[1])
make = lambda i: syn.make_replay(8, 60, seed=10 + i, stride=5).ranges
finals, local_poses, (lo, hi) = slam_dist.replay_sharded(make, n_traj, -3.14159, 3.14159, runner=runner, backend="gloo")
# shared-map case: every rank casts its trajectories into private counters, then all_reduce
g = co.Grid(200, 200)
for i in range(lo, hi):
    co.replay(make(i), -3.14159, 3.14159, g)
ps, ht = slam_dist.all_reduce_counters(g.pass_cnt.copy(), g.hit_cnt.copy())
import torch.distributed as dist
rank = dist.get_rank() if dist.is_initialized() else 0
np.savez(os.path.join(sys.argv[2], "rank%%d.npz" %% rank), finals=finals, lo=lo, hi=hi, ps=ps, ht=ht)
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
'''


def test_shard_range_partitions():
    d = pkg("dist")
    for n in (0, 1, 7, 8, 9, 64):
        for w in (1, 2, 3, 8):
            blocks = [d.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(w - 1))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world,n_traj", [(2, 4), (2, 5), (3, 4)])
def test_gloo_sharded_replay(tmp_path, world, n_traj):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "pkg": PKG})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + (os.getpid() + world * 7 + n_traj) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(n_traj), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # single-process truth
    from oracle import c_oracle as co
    syn = pkg("synthetic")
    truth, g = [], co.Grid(200, 200)
    for i in range(n_traj):
        rr = syn.make_replay(8, 60, seed=10 + i, stride=5).ranges
        truth.append(co.replay(rr, -3.14159, 3.14159, g)[0][-1])
    truth = np.array(truth)
    covered = []
    for rank in range(world):
        z = np.load(tmp_path / ("rank%d.npz" % rank))
        assert np.array_equal(z["finals"], truth)            # every rank holds every final pose
        assert np.array_equal(z["ps"], g.pass_cnt) and np.array_equal(z["ht"], g.hit_cnt)   # merged map is exact
        covered.append((int(z["lo"]), int(z["hi"])))
    assert covered[0][0] == 0 and covered[-1][1] == n_traj
