"""KITTI-layout scan I/O around the hot path: what the reference's ROS driver does before
and after `mainProcess`, so a mounted dataset runs with one command (tools/run_sequence.py).

  * load_cloud: ros/ros1/InsUtils.hpp:502-537 - the point count comes from the LABEL file (uint32 per point, the
    low 16 bits are the SemanticKITTI id, remapped through label_map :50-85 to the 0..25 training ids; unknown ids
    -> 0), the cloud file holds float32 x,y,z,intensity; a missing label file means "end of sequence" (empty
    result), a cloud file shorter than the labels leaves the missing points at the origin (they are cropped by
    min_range later).
  * scan file names: %06d.bin / %06d.label (ros/ros1/semSLAM.cpp:158-163).
  * pose file: one line per scan, the 3x4 pose matrix row-major, fixed notation with 16 decimals, single spaces
    (semSLAM.cpp:231-241).
  * MulRAN timestamps: floor(i / 64) / 1024 for the 64 x 1024 Ouster layout, all ones when the scan has another
    size (semSLAM.cpp:143-151, :176-178).
"""
import os

import numpy as np

from . import synth
from .trajectory import as_pose_array


def scan_paths(lidar_dir, label_dir, index):
    return os.path.join(lidar_dir, f"{index:06d}.bin"), os.path.join(label_dir, f"{index:06d}.label")


def load_cloud(bin_path, label_path, remap=True):
    """-> (xyzi float32 (n, 4), raw uint32 (n,), label int32 (n,)); n == 0 when the label file is missing."""
    if not os.path.isfile(label_path):
        return np.zeros((0, 4), np.float32), np.zeros(0, np.uint32), np.zeros(0, np.int32)
    raw = np.fromfile(label_path, dtype=np.uint32)
    n = len(raw)
    xyzi = np.zeros((n, 4), dtype=np.float32)
    if os.path.isfile(bin_path):
        vals = np.fromfile(bin_path, dtype=np.float32, count=4 * n)
        xyzi.reshape(-1)[:len(vals)] = vals
    label = synth.remap_labels(raw) if remap else raw.astype(np.int32)
    return xyzi, raw, label


def save_cloud(bin_path, label_path, xyzi, raw_label):
    np.ascontiguousarray(xyzi, dtype=np.float32).tofile(bin_path)
    np.ascontiguousarray(raw_label, dtype=np.uint32).tofile(label_path)


def mulran_timestamps(n):
    if n == 64 * 1024:
        return synth.mulran_timestamps(n)
    return np.ones(n, dtype=np.float64)


def pose_line(pose):
    """One line of the odometry result file for a 4x4 pose (or pose7)."""
    M = as_pose_array(np.asarray(pose)[None])[0]
    return " ".join(f"{M[r, c]:.16f}" for r in range(3) for c in range(4))


def write_kitti_poses(path, poses, append=False):
    with open(path, "a" if append else "w") as f:
        for p in as_pose_array(poses):
            f.write(pose_line(p) + "\n")


def read_kitti_poses(path):
    rows = [[float(v) for v in line.split()] for line in open(path) if line.strip()]
    return as_pose_array(np.asarray(rows, dtype=np.float64).reshape(-1, 12))


def read_kitti_calib(path):
    """calib.txt -> {key: 4x4}; the LiDAR->camera transform is calib["Tr"]."""
    out = {}
    for line in open(path):
        if ":" not in line:
            continue
        key, content = line.strip().split(":", 1)
        v = [float(x) for x in content.split()]
        T = np.eye(4)
        T[:3, :] = np.asarray(v[:12]).reshape(3, 4)
        out[key] = T
    return out
