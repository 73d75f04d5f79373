#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

PARITY UNPINNED: the reference (nubot-nudt/SG-SLAM) ships no tests, golden vectors or recorded
outputs for this path, and its C++ cannot be built in this image (Eigen, Sophus, TBB, tsl::robin_map
are absent), so these vectors come from the CPU oracle (oracle/, a restatement of the reference's
algorithm), not from the reference itself.  They pin (a) known answers that follow directly from the
reference's formulas and can be checked by hand (voxel hash, truncating voxel keys, semantic weights)
and (b) the oracle's output on a small seeded sequence, so that neither the oracle nor the HIP path
can drift unnoticed.  Re-run only on purpose:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_py as O  # noqa: E402
from sg_slam_amd import synth  # noqa: E402

FRAMES = 8
SHAPE = "mini"


def main():
    # (a) hand-checkable known answers from the reference's formulas
    keys = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, -1], [12, -7, 3], [1000, -1000, 17]], dtype=np.int32)
    # C/Coreutils.h:40-45: ((1 << 20) - 1) & (x * 73856093 ^ y * 19349663 ^ z * 83492791) on uint32
    h = [((x * 73856093) & 0xFFFFFFFF ^ (y * 19349663) & 0xFFFFFFFF ^ (z * 83492791) & 0xFFFFFFFF) & 0xFFFFF for x, y, z in keys.tolist()]
    pts = np.array([[0.49, -0.49, 0.0], [-0.51, 0.51, 1.0], [2.999999, -3.000001, 0.25], [-0.0, 7.5, -7.5]])
    # C/Coreutils.h:33-37: static_cast<int>(coordinate / voxel_size) truncates toward zero
    trunc_keys = np.trunc(pts / 0.5).astype(np.int32)
    np.savez(os.path.join(HERE, "known_answers.npz"), hash_keys=keys, hash_values=np.array(h, dtype=np.uint32),
             key_points=pts, key_voxel=np.array(0.5), key_values=trunc_keys)

    # (b) the oracle on a small seeded sequence
    slam = O.Slam(O.kitti_config())
    poses, counts, boxes_all, clouds = [], [], [], {}
    for k in range(FRAMES):
        xyzi, raw, lab = synth.scan(k, SHAPE)
        r = slam.process(xyzi[:, :3].astype(np.float64), lab)
        poses.append(r["pose"])
        counts.append([r["n_in"], r["n_cluster"], r["n_ds"], r["n_src"], r["iterations"], r["map_voxels"], r["n_boxes"]])
        boxes_all.append(slam.boxes())
        if k == FRAMES - 1:
            for which in (1, 2, 3):
                x, l = slam.cloud(which)
                clouds[f"cloud{which}_xyz"] = x
                clouds[f"cloud{which}_label"] = l
    np.savez_compressed(os.path.join(HERE, "mini_sequence.npz"), frames=np.array(FRAMES), poses=np.array(poses),
                        counts=np.array(counts, dtype=np.int64), last_boxes=boxes_all[-1],
                        n_boxes=np.array([len(b) for b in boxes_all]), **clouds)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
