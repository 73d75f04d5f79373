"""world_size-2 rehearsal of the sequence-parallel runner on CPU (gloo): every rank drives its own scan
stream (here through the oracle, there is no GPU), timing is the max over ranks, no data collective."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from sg_slam_amd import synth  # noqa: E402
from sg_slam_amd.runner import Dist, assign_sequences, timed_steps  # noqa: E402

dist = Dist(backend="gloo")
O.set_threads(1)
slam = O.Slam(O.kitti_config(enable_graph=0))
poses = []
delay = 0.02 * dist.rank  # make rank 1 measurably slower: the reported time must be ITS time


def step(i):
    xyzi, raw, lab = synth.scan(i, "mini", seed=synth.SEED + dist.rank)
    poses.append(slam.process(xyzi[:, :3].astype(np.float64), lab)["pose"])
    time.sleep(delay)


K = 4
t0 = time.perf_counter()
elapsed, job_steps = timed_steps(dist, step, K)
own = time.perf_counter() - t0
mine = assign_sequences([4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201], dist.world)[dist.rank]
out = {"rank": dist.rank, "world": dist.world, "elapsed": elapsed, "own": own, "job_steps": job_steps, "n_poses": len(poses),
       "last_t": [float(v) for v in poses[-1][4:]], "mine": mine}
with open(os.path.join(os.environ["SGS_TEST_OUT"], f"rank{dist.rank}.json"), "w") as f:
    json.dump(out, f)
dist.close()
