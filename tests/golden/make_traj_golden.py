"""Generates tests/golden/traj_eval.npz + traj_eval.json by running the REFERENCE's evaluation Python
(/root/reference/eval/eval_traj_utils.py: absolute_error, relative_error, apply_kitti_format_calib) on seeded
trajectory pairs.  Run in the build container only (the reference does not travel); the outputs are data.

    python tests/golden/make_traj_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/eval")
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import eval_traj_utils as R  # noqa: E402


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def drive(n, step, yaw_rate, rng, wobble=0.0):
    """A car-like ground-truth trajectory: forward `step` m per frame, slowly varying yaw, small pitch/roll."""
    T = np.eye(4)
    out = []
    for i in range(n):
        out.append(T.copy())
        d = np.eye(4)
        d[:3, :3] = rot([0, 0, 1], yaw_rate * (1 + 0.5 * np.sin(i / 40))) @ rot([0, 1, 0], wobble * rng.normal())
        d[:3, 3] = [step, 0, 0]
        T = T @ d
    return np.array(out)


def drifted(gt, rng, t_sigma, r_sigma, bias_yaw=0.0):
    """Estimated trajectory: the ground-truth increments with noise and a yaw bias, re-integrated."""
    T = np.eye(4)
    out = [T.copy()]
    for i in range(1, len(gt)):
        d = np.linalg.inv(gt[i - 1]) @ gt[i]
        e = np.eye(4)
        e[:3, :3] = rot(rng.normal(size=3), r_sigma * rng.normal()) @ rot([0, 0, 1], bias_yaw)
        e[:3, 3] = t_sigma * rng.normal(size=3)
        T = T @ d @ e
        out.append(T.copy())
    return np.array(out)


def main():
    rng = np.random.default_rng(20251031)
    Tr = np.eye(4)  # KITTI-like LiDAR->camera calibration
    Tr[:3, :3] = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], float) @ rot([0.3, 0.1, 0.9], 0.013)
    Tr[:3, 3] = [-0.0119, -0.0540, -0.2922]
    cases = {}
    arrays = {"Tr": Tr}

    def add(name, gt, est, calib=None):
        arrays[name + "_gt"] = gt
        arrays[name + "_est"] = est
        est_eval = np.array(R.apply_kitti_format_calib(list(est), calib)) if calib is not None else est
        rot_rmse, tran_rmse, align = R.absolute_error(gt, est_eval)
        rot_raw, tran_raw, _ = R.absolute_error(gt, est_eval, align_on=False)
        drift_t, drift_r = R.relative_error(gt, est_eval)
        cases[name] = {"calib": calib is not None, "rot_rmse_deg": float(rot_rmse), "tran_rmse_m": float(tran_rmse),
                       "align": np.asarray(align).tolist(), "rot_rmse_noalign_deg": float(rot_raw),
                       "tran_rmse_noalign_m": float(tran_raw), "drift_t_percent": float(drift_t), "drift_r_deg_per_m": float(drift_r),
                       "dist_last": float(R.trajectory_distances(gt)[-1])}

    gt = drive(400, 1.0, 0.004, rng, 0.001)
    add("ring400", gt, drifted(gt, rng, 0.01, 0.0005, 1e-5))
    gt = drive(1300, 0.9, 0.006, rng, 0.002)
    add("long1300", gt, drifted(gt, rng, 0.02, 0.001, 3e-5))
    gt = drive(60, 0.8, 0.01, rng)
    add("short60", gt, drifted(gt, rng, 0.01, 0.001))             # < 100 m: relative error (0, 0)
    gt = drive(300, 1.2, -0.003, rng, 0.001)
    add("exact300", gt, gt.copy())                                 # zero error
    G = np.eye(4); G[:3, :3] = rot([0.2, -0.4, 1.0], 0.7); G[:3, 3] = [5, -3, 1]
    add("moved300", gt, np.array([G @ p for p in gt]))             # a rigid offset: alignment removes the translation part
    gt = drive(500, 1.0, 0.005, rng, 0.001)
    lidar = np.array([np.linalg.inv(Tr) @ p @ Tr for p in drifted(gt, rng, 0.015, 0.0008, 2e-5)])
    add("calib500", gt, lidar, calib=Tr)                           # estimate in the LiDAR frame, calibrated before scoring
    np.savez_compressed(os.path.join(HERE, "traj_eval.npz"), **arrays)
    with open(os.path.join(HERE, "traj_eval.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_traj_golden.py", "source": "reference eval/eval_traj_utils.py run in the build container",
                   "cases": cases}, f, indent=1)
    for k, v in cases.items():
        print(k, {a: b for a, b in v.items() if a != "align"})


if __name__ == "__main__":
    main()
