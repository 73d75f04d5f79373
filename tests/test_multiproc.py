"""N > 1 path on CPU: two gloo ranks launched the way the driver launches bench.py."""
import json
import os
import subprocess
import sys
import tempfile

from sg_slam_amd.runner import assign_sequences

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assign_sequences_balances_longest_first():
    kitti = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]  # KITTI 00-10 scan counts
    for world in (1, 2, 4, 8):
        parts = assign_sequences(kitti, world)
        assert sorted(i for p in parts for i in p) == list(range(11))
        loads = [sum(kitti[i] for i in p) for p in parts]
        assert max(loads) <= max(max(kitti), sum(kitti) / world * 1.35)
    assert assign_sequences(kitti, 2) == assign_sequences(kitti, 2)  # deterministic


def test_two_gloo_ranks_time_is_max_over_ranks():
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, SGS_TEST_OUT=d, OMP_NUM_THREADS="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29613", os.path.join(ROOT, "tests", "_gloo_worker.py")]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res = [json.load(open(os.path.join(d, f"rank{k}.json"))) for k in range(2)]
    assert all(x["world"] == 2 and x["n_poses"] == 4 for x in res)
    assert res[0]["job_steps"] == res[1]["job_steps"] == 8          # whole-job steps = sum over ranks
    assert abs(res[0]["elapsed"] - res[1]["elapsed"]) < 1e-9          # both ranks report the same (max) time
    assert res[0]["elapsed"] >= max(0.06, res[1]["own"] * 0.5)        # ... and it is the slow rank's
    assert res[0]["last_t"] != res[1]["last_t"]                       # independent streams (different seeds)
    assert sorted(res[0]["mine"] + res[1]["mine"]) == list(range(11))  # sequences dealt without overlap


def test_two_gloo_ranks_key_sharded_map_registers_like_one_map():
    """map sharded by voxel key, per-iteration all-reduce of the 28 Gauss-Newton scalars."""
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, SGS_TEST_OUT=d, OMP_NUM_THREADS="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29617", os.path.join(ROOT, "tests", "_gloo_shard_worker.py")]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res = [json.load(open(os.path.join(d, f"shard{k}.json"))) for k in range(2)]
    a, b = res
    assert a["pose"] == b["pose"] and a["iters"] == b["iters"]           # same sums -> same solve on every rank, bit for bit
    assert a["counts"][0] + b["counts"][0] == a["counts"][1] == a["counts"][2]  # each correspondence found by exactly one rank
    assert min(a["counts"][0], b["counts"][0]) > 0.2 * a["counts"][2]    # and the work is really split
    assert abs(a["iters"] - a["ref_iters"]) <= 1
    assert max(abs(x - y) for x, y in zip(a["pose"], a["ref_pose"])) < 1e-7  # summation order only
    assert max(a["voxels"], b["voxels"]) < a["full_voxels"] <= a["voxels"] + b["voxels"]  # shares + halo
