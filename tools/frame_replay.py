#!/usr/bin/env python3
"""Teacher-forced replay of single frames of a long drive (diagnostic): the oracle runs the drive; at the frames
named on the command line its local map is loaded into the library (same voxels, same in-voxel order) and the
frame's registration is run on both from the same guess and thresholds.
usage: python tools/frame_replay.py 1000 754 761 766 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from helpers import pose_err  # noqa: E402
from sg_slam_amd import api, synth  # noqa: E402

frames = int(sys.argv[1])
want = set(int(a) for a in sys.argv[2:])
slam = O.Slam(O.kitti_config(enable_graph=0))
ctx = api.Context(api.preset("kitti", max_points=1 << 18, map_capacity_log2=17, enable_graph=0))
poses = []
for k in range(frames):
    xyzi, raw, lab = synth.scan(k, "hdl64")
    xyz = xyzi[:, :3].astype(np.float64)
    if k in want:
        keys, cnt, pts = slam.map().dump()
        flat = np.concatenate([pts[i, :cnt[i]] for i in range(len(cnt))])
        ctx.Clear()
        ctx.AddPoints(flat)
        px, pl = O.preprocess(xyz, lab, 100.0, 5.0)
        dx, dl, _, _ = O.voxel_downsample(px, pl, 0.5)
        sx, sl, _, _ = O.voxel_downsample(dx, dl, 1.5)
        q = np.concatenate([sx, sl[:, None].astype(np.float64)], axis=1)
    ro = slam.process(xyz, lab)
    if k in want:
        # the guess and sigma the oracle used: constant-velocity prediction from its pose history
        pred = O.se3_mul(O.se3_inv(poses[-2]), poses[-1])
        guess = O.se3_mul(poses[-1], pred)
        sigma = ro["sigma"]
        pg, itg = ctx.RegisterFrameSemantic(q, guess, 3 * sigma, sigma / 3)
        om = O.Map(1.0, 100.0, 20)
        om.add_points(flat)
        po, info = om.register_frame(q, guess, 3 * sigma, sigma / 3)
        print(k, "lib vs oracle (same map):", pose_err(po, pg), "iters", itg, info["iterations"], "| oracle replay vs its pipeline pose:",
              pose_err(po, ro["pose"]), flush=True)
        # sensitivity of THIS frame: the same registration from a guess moved by 4e-6 m
        g2 = guess.copy(); g2[4] += 4e-6
        po2, info2 = om.register_frame(q, g2, 3 * sigma, sigma / 3)
        print("   oracle, guess moved by 4e-6 m:", pose_err(po, po2), "iters", info2["iterations"], flush=True)
    poses.append(ro["pose"])
