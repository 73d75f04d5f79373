"""One rank of tests/test_gpu_parity.py::test_key_sharded_map_two_gloo_processes_one_gpu: the HIP library's
sharded registration with a gloo all-reduce as the callback, checked against the oracle's single map."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from conftest import make_ctx  # noqa: E402
from sg_slam_amd import synth  # noqa: E402
from sg_slam_amd.runner import Dist  # noqa: E402
from sg_slam_amd.shard import torch_allreduce  # noqa: E402

dist = Dist(backend="gloo")
O.set_threads(1)


def xyzl(x, l):
    return np.concatenate([x, np.asarray(l, dtype=np.float64)[:, None]], axis=1)


ctx = make_ctx()
ctx.ShardSet(dist.rank, dist.world, 2, torch_allreduce)
om = O.Map(1.0, 100.0, 20)
for k in range(3):
    xyzi, raw, lab = synth.scan(k, "small")
    px, pl = O.preprocess(xyzi[:, :3].astype(np.float64), lab, 100.0, 5.0)
    dx, dl, _, _ = O.voxel_downsample(px, pl, 0.5)
    om.update(xyzl(dx, dl), synth.gt_pose(k))
    ctx.Update(xyzl(dx, dl), synth.gt_pose(k))
xyzi, raw, lab = synth.scan(3, "small")
px, pl = O.preprocess(xyzi[:, :3].astype(np.float64), lab, 100.0, 5.0)
dx, dl, _, _ = O.voxel_downsample(px, pl, 0.5)
sx, sl, _, _ = O.voxel_downsample(dx, dl, 1.5)
guess = O.se3_mul(synth.gt_pose(3), O.se3_exp([0.3, 0.05, 0.0, 0, 0, 0.01]))
ref_pose, info = om.register_frame(xyzl(sx, sl), guess, 6.0, 0.667)
pose, iters = ctx.RegisterFrameSemantic(xyzl(sx, sl), guess, 6.0, 0.667)
out = {"rank": dist.rank, "pose": [float(v) for v in pose], "ref_pose": [float(v) for v in ref_pose], "iters": int(iters),
       "ref_iters": int(info["iterations"]), "voxels": int(ctx.map_size()[0]), "full_voxels": int(om.size()[0])}
with open(os.path.join(os.environ["SGS_TEST_OUT"], f"gshard{dist.rank}.json"), "w") as f:
    json.dump(out, f)
ctx.close()
dist.close()
