"""world_size-2 rehearsal of the key-sharded map on CPU (gloo): every rank keeps its share of the voxels
(+halo) in an oracle map, answers the queries whose centre voxel it owns, and the ranks all-reduce the
28 scalars of the Gauss-Newton system per iteration - the data path of sgs_register_frame_sharded with the
oracle standing in for the kernels."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from sg_slam_amd import synth  # noqa: E402
from sg_slam_amd.runner import Dist  # noqa: E402
from sg_slam_amd.shard import shard_owner, shard_stores, torch_allreduce  # noqa: E402

dist = Dist(backend="gloo")
O.set_threads(1)
B = 2  # 4-voxel blocks: many block borders inside the mini scans


def xyzl(x, l):
    return np.concatenate([x, np.asarray(l, dtype=np.float64)[:, None]], axis=1)


full, mine = O.Map(1.0, 100.0, 20), O.Map(1.0, 100.0, 20)
for k in range(3):
    xyzi, raw, lab = synth.scan(k, "mini")
    px, pl = O.preprocess(xyzi[:, :3].astype(np.float64), lab, 100.0, 5.0)
    dx, dl, _, _ = O.voxel_downsample(px, pl, 0.5)
    w = O.se3_act(synth.gt_pose(k), dx)
    keep = shard_stores(O.voxel_keys(w, 1.0), B, dist.world, dist.rank)
    full.add_points(xyzl(w, dl))
    mine.add_points(xyzl(w[keep], dl[keep]))

xyzi, raw, lab = synth.scan(3, "mini")
px, pl = O.preprocess(xyzi[:, :3].astype(np.float64), lab, 100.0, 5.0)
dx, dl, _, _ = O.voxel_downsample(px, pl, 0.5)
sx, sl, _, _ = O.voxel_downsample(dx, dl, 1.5)
guess = O.se3_mul(synth.gt_pose(3), O.se3_exp([0.3, 0.05, 0.0, 0, 0, 0.01]))
ref_pose, info = full.register_frame(xyzl(sx, sl), guess, 6.0, 0.667)

T = guess
iters, first_counts = 0, None
while iters < 500:
    w = O.se3_act(T, sx)
    own = shard_owner(O.voxel_keys(w, 1.0), B, dist.world) == dist.rank
    sys27, cnt = mine.build_linear_system(xyzl(w[own], sl[own]), 6.0, 0.667)
    buf = np.concatenate([sys27, [float(cnt)]])
    torch_allreduce(buf)
    if first_counts is None:
        _, full_cnt = full.build_linear_system(xyzl(w, sl), 6.0, 0.667)
        first_counts = [int(cnt), int(buf[27]), int(full_cnt)]
    A = np.zeros((6, 6))
    A[np.triu_indices(6)] = buf[:21]
    A = A + A.T - np.diag(np.diag(A))
    x = O.ldlt6_solve(A, -buf[21:27])
    T = O.se3_mul(O.se3_exp(x), T)
    iters += 1
    if np.linalg.norm(x) < 1e-4:
        break

out = {"rank": dist.rank, "world": dist.world, "pose": [float(v) for v in T], "ref_pose": [float(v) for v in ref_pose],
       "iters": iters, "ref_iters": int(info["iterations"]), "counts": first_counts,
       "voxels": int(mine.size()[0]), "full_voxels": int(full.size()[0])}
with open(os.path.join(os.environ["SGS_TEST_OUT"], f"shard{dist.rank}.json"), "w") as f:
    json.dump(out, f)
dist.close()
