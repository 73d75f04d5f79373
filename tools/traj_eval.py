"""Trajectory evaluation with the reference's command line (eval/traj_eval.py:25-32) and output lines.

    python tools/traj_eval.py --gt_file poses/00.txt --pred_file odom_00.txt --calib_file calib.txt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sg_slam_amd import kitti_io, trajectory  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="KITTI trajectory evaluation")
    ap.add_argument("--gt_file", default="./ground_truth_pose")
    ap.add_argument("--pred_file", default="./pred_pose")
    ap.add_argument("--calib_file", default="calib.txt", help="KITTI calib.txt with a Tr: line; 'none' = poses already in the gt frame")
    a = ap.parse_args()
    Tr = None if a.calib_file.lower() == "none" else kitti_io.read_kitti_calib(a.calib_file)["Tr"]
    out = trajectory.evaluate(kitti_io.read_kitti_poses(a.gt_file), kitti_io.read_kitti_poses(a.pred_file), Tr)
    print("Average Translation Error       (%):", out["Average Translation Error (%)"])
    print("Average Rotational Error (deg/100m):", out["Average Rotational Error (deg/100m)"])
    print("Absoulte Trajectory Error       (m):", out["Absolute Trajectory Error (m)"])
    print("Absoulte Rotation Error       (deg):", out["Absolute Rotation Error (deg)"])


if __name__ == "__main__":
    main()
