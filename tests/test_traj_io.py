"""Row f4 of SURVEY.md 8f: trajectory metrics against numbers produced by the reference's own evaluation
Python (tests/golden/traj_eval.json, made by tests/golden/make_traj_golden.py), and the KITTI-layout I/O."""
import json
import os

import numpy as np
import pytest

from sg_slam_amd import kitti_io, synth, trajectory

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "traj_eval.json")))["cases"]
ARR = np.load(os.path.join(GOLD, "traj_eval.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_metrics_match_reference_eval(name):
    want = CASES[name]
    gt, est = ARR[name + "_gt"], ARR[name + "_est"]
    if want["calib"]:
        est = trajectory.apply_calibration(est, ARR["Tr"])
    rot, tran, A = trajectory.absolute_error(gt, est)
    # the angle comes from arccos(1 - eps): rounding-level differences are amplified to ~1e-6 deg near zero
    assert tran == pytest.approx(want["tran_rmse_m"], rel=1e-9, abs=1e-11)
    assert rot == pytest.approx(want["rot_rmse_deg"], rel=1e-7, abs=5e-6)
    assert np.allclose(A, np.array(want["align"]), rtol=0, atol=1e-9)
    rot0, tran0, A0 = trajectory.absolute_error(gt, est, align_on=False)
    assert np.array_equal(A0, np.eye(4))
    assert tran0 == pytest.approx(want["tran_rmse_noalign_m"], rel=1e-12, abs=1e-13)
    assert rot0 == pytest.approx(want["rot_rmse_noalign_deg"], rel=1e-7, abs=5e-6)
    dt, dr = trajectory.relative_error(gt, est)
    assert dt == pytest.approx(want["drift_t_percent"], rel=1e-9, abs=1e-12)
    assert dr == pytest.approx(want["drift_r_deg_per_m"], rel=1e-7, abs=1e-8)
    assert trajectory.trajectory_distances(gt)[-1] == pytest.approx(want["dist_last"], rel=1e-13)


def test_relative_error_needs_100_m():
    assert CASES["short60"]["drift_t_percent"] == 0.0
    assert trajectory.relative_error(ARR["short60_gt"], ARR["short60_est"]) == (0.0, 0.0)


def test_evaluate_reports_the_four_numbers_of_traj_eval():
    out = trajectory.evaluate(ARR["calib500_gt"], ARR["calib500_est"], ARR["Tr"])
    w = CASES["calib500"]
    assert out["Absolute Trajectory Error (m)"] == pytest.approx(w["tran_rmse_m"], rel=1e-9)
    assert out["Average Translation Error (%)"] == pytest.approx(w["drift_t_percent"], rel=1e-9)
    assert out["Average Rotational Error (deg/100m)"] == pytest.approx(100 * w["drift_r_deg_per_m"], rel=1e-7)


def test_pose7_and_kitti_rows_are_accepted():
    gt = ARR["ring400_gt"]
    rows = gt[:, :3, :].reshape(-1, 12)
    assert np.array_equal(trajectory.as_pose_array(rows), gt)
    p7 = np.array([synth.gt_pose(k) for k in range(5)])
    M = trajectory.pose7_to_matrix(p7)
    assert np.allclose(M[:, :3, :3] @ M[:, :3, :3].transpose(0, 2, 1), np.eye(3), atol=1e-14)
    assert np.array_equal(M[:, :3, 3], p7[:, 4:])
    # rotation of e_x by the quaternion, written out
    x, y, z, w = p7[3, :4]
    assert np.allclose(M[3, :3, 0], [1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)])


def test_pose_file_format_and_round_trip(tmp_path):
    # semSLAM.cpp:235-240: fixed notation, 16 decimals, 12 numbers separated by single spaces
    M = np.eye(4); M[0, 3] = 1.5; M[1, 3] = -0.25; M[2, 2] = 1.0
    line = kitti_io.pose_line(M)
    assert line == ("1.0000000000000000 0.0000000000000000 0.0000000000000000 1.5000000000000000 "
                    "0.0000000000000000 1.0000000000000000 0.0000000000000000 -0.2500000000000000 "
                    "0.0000000000000000 0.0000000000000000 1.0000000000000000 0.0000000000000000")
    poses = ARR["ring400_est"]
    p = tmp_path / "odom.txt"
    kitti_io.write_kitti_poses(p, poses[:200])
    kitti_io.write_kitti_poses(p, poses[200:], append=True)  # the reference appends scan by scan
    back = kitti_io.read_kitti_poses(p)
    assert back.shape == poses.shape and np.abs(back - poses).max() < 1e-15 * 400
    assert all(len(l.split(" ")) == 12 for l in open(p).read().splitlines())


def test_calib_file(tmp_path):
    Tr = ARR["Tr"]
    f = tmp_path / "calib.txt"
    f.write_text("P0: " + " ".join(["1.0"] * 12) + "\nTr: " + " ".join(repr(float(v)) for v in Tr[:3].reshape(-1)) + "\n")
    c = kitti_io.read_kitti_calib(f)
    assert set(c) == {"P0", "Tr"} and np.array_equal(c["Tr"], Tr)
    assert np.array_equal(c["P0"][3], [0, 0, 0, 1])


def test_load_cloud_follows_the_reference_loader(tmp_path):
    xyzi, raw, lab = synth.scan(3, "mini")
    raw = raw | (np.arange(len(raw), dtype=np.uint32) << 16)  # instance ids in the upper half must be ignored
    b, l = kitti_io.scan_paths(tmp_path, tmp_path, 3)
    assert b.endswith("000003.bin") and l.endswith("000003.label")
    kitti_io.save_cloud(b, l, xyzi, raw)
    x2, r2, l2 = kitti_io.load_cloud(b, l)
    assert np.array_equal(x2, xyzi) and np.array_equal(r2, raw) and np.array_equal(l2, lab)
    assert l2.dtype == np.int32 and l2.min() >= 0 and l2.max() <= 25
    # no label file -> end of sequence (InsUtils.hpp:504-509)
    x3, r3, l3 = kitti_io.load_cloud(b, str(tmp_path / "000004.label"))
    assert len(x3) == len(r3) == len(l3) == 0
    # the point count is the label file's; a short cloud file leaves zeros (vector value-initialisation, :516-517)
    np.ascontiguousarray(xyzi[:10]).tofile(b)
    x4, _, l4 = kitti_io.load_cloud(b, l)
    assert len(x4) == len(raw) and np.array_equal(x4[:10], xyzi[:10]) and not x4[10:].any() and np.array_equal(l4, lab)
    # ids the map does not know become class 0 (std::map::operator[] default)
    np.array([7, 10, 0xFFFF0000 | 252, 99], dtype=np.uint32).tofile(l)
    _, _, l5 = kitti_io.load_cloud(b, l)
    assert l5.tolist() == [0, synth.LABEL_MAP[10], synth.LABEL_MAP[252], synth.LABEL_MAP.get(99, 0)]


def test_mulran_timestamps():
    t = kitti_io.mulran_timestamps(64 * 1024)
    assert t[0] == 0.0 and t[63] == 0.0 and t[64] == 1 / 1024 and t[-1] == 1023 / 1024
    assert np.array_equal(kitti_io.mulran_timestamps(1000), np.ones(1000))  # semSLAM.cpp:176-178
