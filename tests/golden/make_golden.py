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
    # (c) rows beyond the hot path: scan descriptor of the last frame, a loop-pair refinement, SolveSVD /
    # seeded RANSAC on a fixed point set
    cx, cl = slam.cloud(1)
    desc = O.scan_descriptor(boxes_all[-1], cx, cl)

    def ds_cloud(k):
        xyzi, raw, lab = synth.scan(k, SHAPE)
        px, pl = O.preprocess(xyzi[:, :3].astype(np.float64), lab, 100.0, 5.0)
        dx, dl, _, _ = O.voxel_downsample(px, pl, 0.5)
        return np.concatenate([dx, dl[:, None].astype(np.float64)], axis=1)
    A, B = ds_cloud(5), ds_cloud(3)
    init = O.se3_mul(O.se3_mul(O.se3_inv(synth.gt_pose(3)), synth.gt_pose(5)), O.se3_exp([0.1, -0.05, 0.0, 0, 0, 0.005]))
    ficp_pose, ficp_info = O.fast_icp(A, B, init, 1.0, 5, 0.5)
    rng = np.random.default_rng(20251031)
    T = O.se3_exp([0.5, -1.0, 0.3, 0.1, -0.2, 0.4])
    nodes1 = rng.uniform(-40, 40, size=(24, 3)); nodes2 = O.se3_act(T, nodes1)
    idx = rng.permutation(24)[:10]
    m1, m2 = nodes1[idx].copy(), nodes2[idx].copy()
    m2[:3] += rng.normal(size=(3, 3)) * 3
    R, t, score, best = O.ransac_alignment(m1, m2, nodes1, nodes2, 500, 7)
    Rs, ts = O.solve_svd(m1[3:], m2[3:])
    np.savez_compressed(os.path.join(HERE, "next_rows.npz"), scan_descriptor=desc, ficp_a=A, ficp_b=B, ficp_init=init, ficp_pose=ficp_pose,
                        ficp_iterations=np.array(ficp_info["iterations"]), nodes1=nodes1, nodes2=nodes2, match1=m1, match2=m2,
                        ransac_R=R, ransac_t=t, ransac_score=np.array(score), ransac_best=np.array(best), svd_R=Rs, svd_t=ts,
                        svd_truth=T)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
