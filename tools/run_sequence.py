"""Run a KITTI-layout sequence (NNNNNN.bin + NNNNNN.label) through the HIP odometry path and write the pose
file the reference's driver writes (ros/ros1/semSLAM.cpp:154-241), optionally scoring it.

    python tools/run_sequence.py --lidar seq/00/velodyne --labels seq/00/labels --dataset kitti --out odom_00.txt \\
           [--gt poses/00.txt --calib seq/00/calib.txt]
    python tools/run_sequence.py --make-synthetic /tmp/synth --frames 60 --shape hdl64     # writes scans + gt poses
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from sg_slam_amd import kitti_io, synth, trajectory  # noqa: E402


def make_synthetic(root, frames, shape):
    os.makedirs(os.path.join(root, "velodyne"), exist_ok=True)
    os.makedirs(os.path.join(root, "labels"), exist_ok=True)
    for k in range(frames):
        xyzi, raw, _ = synth.scan(k, shape)
        kitti_io.save_cloud(*kitti_io.scan_paths(os.path.join(root, "velodyne"), os.path.join(root, "labels"), k), xyzi, raw)
    kitti_io.write_kitti_poses(os.path.join(root, "poses.txt"), np.array([synth.gt_pose(k) for k in range(frames)]))


def run(lidar, labels, dataset, out, max_scans=None, preset="kitti", start=0, step=None):
    """-> (poses (n,4,4), seconds spent inside the per-scan call).  `step(xyzi, label, timestamps) -> pose7`
    defaults to the HIP library's sgs_process_scan_f32 (tests pass another engine to compare trajectories)."""
    if step is None:
        from sg_slam_amd import api
        slam = api.Context(api.preset(preset, max_points=1 << 21))
        step = lambda xyzi, lab, ts: slam.process_scan_f32(xyzi, lab, ts, dataset)["pose"]  # noqa: E731
    if out and os.path.exists(out):
        os.remove(out)  # the driver appends line by line
    poses, busy, k = [], 0.0, start
    while max_scans is None or len(poses) < max_scans:
        xyzi, _, lab = kitti_io.load_cloud(*kitti_io.scan_paths(lidar, labels, k))
        if len(xyzi) == 0:
            break  # "No more data" (semSLAM.cpp:166-170)
        ts = kitti_io.mulran_timestamps(len(xyzi)) if dataset == "mulran" else None
        t0 = time.perf_counter()
        p = step(xyzi, lab, ts)
        busy += time.perf_counter() - t0
        poses.append(p)
        if out:
            kitti_io.write_kitti_poses(out, [p], append=True)
        k += 1
    return trajectory.as_pose_array(np.array(poses)) if poses else np.zeros((0, 4, 4)), busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-synthetic")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--shape", default="hdl64")
    ap.add_argument("--lidar"); ap.add_argument("--labels")
    ap.add_argument("--dataset", default="", help="kitti (angle correction), mulran (deskew timestamps) or empty")
    ap.add_argument("--out", default="odom_result.txt")
    ap.add_argument("--gt"); ap.add_argument("--calib")
    ap.add_argument("--max-scans", type=int)
    a = ap.parse_args()
    if a.make_synthetic:
        make_synthetic(a.make_synthetic, a.frames, a.shape)
        if not a.lidar:
            a.lidar, a.labels = os.path.join(a.make_synthetic, "velodyne"), os.path.join(a.make_synthetic, "labels")
            a.gt = a.gt or os.path.join(a.make_synthetic, "poses.txt")
    poses, busy = run(a.lidar, a.labels, a.dataset, a.out, a.max_scans)
    rep = {"scans": len(poses), "scans_per_s_in_call": len(poses) / busy if busy else None, "pose_file": a.out}
    if a.gt and len(poses):
        gt = kitti_io.read_kitti_poses(a.gt)[:len(poses)]
        Tr = kitti_io.read_kitti_calib(a.calib)["Tr"] if a.calib else None
        rep.update(trajectory.evaluate(gt, poses, Tr))
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
