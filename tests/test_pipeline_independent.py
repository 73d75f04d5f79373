"""An independent restatement of the per-scan odometry (SemGraphSLAM::mainProcess without the graph half) in plain
numpy / scipy, written from the reference's sources and sharing no code with the C++ oracle, run beside the oracle on
small scans.  The oracle is "parity unpinned" (the reference ships no vectors and cannot be built here); this does not
pin it to the reference, but it does mean two restatements by different means - dictionaries, numpy.linalg.solve,
scipy's matrix exponential and rotation log on one side, a robin-hood table, a hand-written LDLT and closed-form SE3 maps
on the other - agree on every intermediate size, iteration count and pose.

Followed here (paths under cpp/semgraph_slam/): pipeline/SemGraphSLAM.cpp:16-166 (order of the steps, prediction model,
HasMoved, thresholds 3 sigma and sigma / 3), frontend/Preprocessing.cpp:70-125, frontend/VoxelHashMap.cpp:51-186,
frontend/Registration.cpp:50-177 (incl. the INCREMENTAL transform of the source cloud, :165), frontend/Registration.hpp:
21-56, frontend/Threshold.cpp:29-50."""
import numpy as np
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation

import oracle_py as O
from helpers import pose_matrix, robin_order, trunc_key
from sg_slam_amd import synth

W_SEM = {16: 1.2, 18: 1.2, 19: 1.2, **{k: 0.0 for k in range(20, 26)}}  # Registration.hpp:21-56; everything else 1.0


def downsample(xyz, lab, voxel):
    """Preprocessing.cpp:70-94: first point per voxel, emitted in the robin_map's bucket order."""
    idx = robin_order(trunc_key(xyz, voxel))
    return xyz[idx], lab[idx]


class PyMap:
    """VoxelHashMap.cpp:51-186 on a dict (voxel -> list of (x, y, z, label)), pure prune predicate."""

    def __init__(self, voxel, max_dist, cap):
        self.voxel, self.max_dist, self.cap, self.m = voxel, max_dist, cap, {}

    def add(self, pts4):
        for p in pts4:
            k = tuple(int(v) for v in np.trunc(p[:3] / self.voxel))
            b = self.m.setdefault(k, [])
            if len(b) < self.cap:
                b.append(p.copy())

    def prune(self, origin):
        d2 = self.max_dist * self.max_dist
        self.m = {k: b for k, b in self.m.items() if not np.sum((b[0][:3] - origin) ** 2) > d2}

    def correspondences(self, src4, max_dist):
        s_out, t_out = [], []
        for p in src4:
            kx, ky, kz = (int(v) for v in np.trunc(p[:3] / self.voxel))
            cand = []
            for i in (kx - 1, kx, kx + 1):
                for j in (ky - 1, ky, ky + 1):
                    for k in (kz - 1, kz, kz + 1):
                        cand.extend(self.m.get((i, j, k), ()))
            if not cand:
                continue  # the reference's uninitialised neighbour: defined as "no correspondence" 
            c = np.array(cand)
            d2 = np.sum((c[:, :3] - p[:3]) ** 2, axis=1)
            best = c[int(np.argmin(d2))]  # argmin returns the FIRST minimum: the strict '<' of :87
            if np.linalg.norm(best[:3] - p[:3]) < max_dist:
                s_out.append(p)
                t_out.append(best)
        return np.array(s_out).reshape(-1, 4), np.array(t_out).reshape(-1, 4)


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp(x):
    X = np.zeros((4, 4))
    X[:3, :3] = hat(x[3:])
    X[:3, 3] = x[:3]
    return expm(X)


def align_clouds(src4, tgt4, th):
    """Registration.cpp:71-113."""
    JTJ, JTr = np.zeros((6, 6)), np.zeros(6)
    for s, t in zip(src4, tgt4):
        r = s[:3] - t[:3]
        J = np.hstack([np.eye(3), -hat(s[:3])])
        w = th * th / (th + r @ r) ** 2
        JTJ += J.T @ (w * J)
        JTr += J.T @ (w * r) * W_SEM.get(int(s[3]), 1.0)
    if not JTJ.any():
        return np.eye(4), np.zeros(6)  # Eigen's LDLT solves the all-zero system to x = 0
    x = np.linalg.solve(JTJ, -JTr)
    return se3_exp(x), x


def register(frame4, pmap, guess, max_corr, kernel):
    """Registration.cpp:147-177."""
    if not pmap.m:
        return guess, 0
    src = frame4.copy()
    src[:, :3] = src[:, :3] @ guess[:3, :3].T + guess[:3, 3]
    T_icp = np.eye(4)
    it = 0
    for it in range(1, 501):
        s, t = pmap.correspondences(src, max_corr)
        est, x = align_clouds(s, t, kernel)
        src[:, :3] = src[:, :3] @ est[:3, :3].T + est[:3, 3]
        T_icp = est @ T_icp
        if np.linalg.norm(x) < 1e-4:  # |log(exp(x))| = |x|
            break
    return T_icp @ guess, it


class PySlam:
    def __init__(self, voxel=1.0, max_range=100.0, min_range=5.0, cap=20, initial_threshold=2.0, min_motion_th=0.1):
        self.voxel, self.max_range, self.min_range = voxel, max_range, min_range
        self.initial_threshold, self.min_motion_th = initial_threshold, min_motion_th
        self.map = PyMap(voxel, max_range, cap)
        self.poses, self.deviation, self.sse2, self.samples = [], np.eye(4), 0.0, 0

    def threshold(self):
        moved = bool(self.poses) and np.linalg.norm((np.linalg.inv(self.poses[0]) @ self.poses[-1])[:3, 3]) > 5.0 * self.min_motion_th
        if not moved:
            return self.initial_threshold
        theta = Rotation.from_matrix(self.deviation[:3, :3]).magnitude()
        err = np.linalg.norm(self.deviation[:3, 3]) + 2.0 * self.max_range * np.sin(theta / 2.0)
        if err > self.min_motion_th:
            self.sse2 += err * err
            self.samples += 1
        return self.initial_threshold if self.samples < 1 else np.sqrt(self.sse2 / self.samples)

    def process(self, xyz, lab, timestamps=None, kitti=False, deskew=False):
        self._raw = xyz
        if deskew and timestamps is not None and len(self.poses) > 2:
            # Deskew.cpp:37-48: exp((t_i - 0.5) * log(T0^-1 T1)) * p_i with the last two poses
            xi = np.real(logm(np.linalg.inv(self.poses[-2]) @ self.poses[-1]))
            ts, first = np.unique(timestamps, return_inverse=True)
            motions = np.array([expm((t - 0.5) * xi) for t in ts])[first]
            xyz = np.einsum("nij,nj->ni", motions[:, :3, :3], xyz) + motions[:, :3, 3]
        if kitti:
            # Preprocessing.cpp:127-137 on the RAW frame (SemGraphSLAM.cpp:30-33 overwrites any deskew result):
            # rotation by 0.205 deg about normalize(p x z)
            raw_xyz = self._raw
            ax = np.cross(raw_xyz, [0.0, 0.0, 1.0])
            nrm_ax = np.linalg.norm(ax, axis=1, keepdims=True)
            ax = np.divide(ax, nrm_ax, out=np.zeros_like(ax), where=nrm_ax > 0)
            xyz = Rotation.from_rotvec(ax * (0.205 * np.pi / 180.0)).apply(raw_xyz)
        nrm = np.sqrt(np.sum(xyz * xyz, axis=1))
        keep = (nrm < self.max_range) & (nrm > self.min_range) & (lab < 20)
        cx, cl = xyz[keep], lab[keep]
        clu = downsample(cx, cl, 0.2)
        dx, dl = downsample(cx, cl, 0.5 * self.voxel)
        sx, sl = downsample(dx, dl, 1.5 * self.voxel)
        sigma = self.threshold()
        pred = np.linalg.inv(self.poses[-2]) @ self.poses[-1] if len(self.poses) >= 2 else np.eye(4)
        guess = (self.poses[-1] if self.poses else np.eye(4)) @ pred
        pose, it = register(np.column_stack([sx, sl.astype(float)]), self.map, guess, 3.0 * sigma, sigma / 3.0)
        self.deviation = np.linalg.inv(guess) @ pose
        world = np.column_stack([dx @ pose[:3, :3].T + pose[:3, 3], dl.astype(float)])
        self.map.add(world)
        self.map.prune(pose[:3, 3])
        self.poses.append(pose)
        return dict(n_in=len(cx), n_cluster=len(clu[0]), n_ds=len(dx), n_src=len(sx), iterations=it, sigma=sigma, pose=pose,
                    map_voxels=len(self.map.m))


def test_numpy_pipeline_and_oracle_agree_scan_by_scan():
    py, orc = PySlam(), O.Slam(O.kitti_config(enable_graph=0))
    worst_t = worst_r = 0.0
    for k in range(10):
        xyzi, _, lab = synth.scan(k, "mini")
        xyz = xyzi[:, :3].astype(np.float64)
        rp, ro = py.process(xyz, lab), orc.process(xyz, lab)
        for a in ("n_in", "n_cluster", "n_ds", "n_src", "map_voxels", "iterations"):
            assert rp[a] == ro[a], f"scan {k}: {a} {rp[a]} != {ro[a]}"
        assert abs(rp["sigma"] - ro["sigma"]) < 1e-9, k
        D = np.linalg.inv(pose_matrix(ro["pose"])) @ rp["pose"]
        worst_t = max(worst_t, np.linalg.norm(D[:3, 3]))
        worst_r = max(worst_r, Rotation.from_matrix(D[:3, :3]).magnitude())
    assert ro["iterations"] > 0 and ro["n_src"] > 50
    assert worst_t < 1e-9 and worst_r < 1e-10, (worst_t, worst_r)
    # the maps hold the same points in the same in-voxel order
    keys, cnt, pts = orc.map().dump()
    assert len(keys) == len(py.map.m)
    for key, c, p in zip(keys.tolist(), cnt, pts):
        b = np.array(py.map.m[tuple(key)])
        assert len(b) == c and np.array_equal(b[:, 3], p[:c, 3]) and np.abs(b[:, :3] - p[:c, :3]).max() < 1e-8


def test_numpy_pipeline_kitti_correction_and_deskew_variants():
    """The same comparison with CorrectKITTIScan on (dataset == "kitti") and, separately, with DeSkewScan on
    (MulRAN-style per-column timestamps); coordinates now pass through scipy's Rodrigues / matrix log + exp on one side
    and the oracle's closed forms on the other, so keys of points on a voxel face could differ - none does here."""
    for mode in ("kitti", "deskew"):
        py = PySlam()
        orc = O.Slam(O.kitti_config(enable_graph=0, deskew=1 if mode == "deskew" else 0))
        for k in range(7):
            xyzi, _, lab = synth.scan(k, "mini")
            xyz = xyzi[:, :3].astype(np.float64)
            ts = synth.mulran_timestamps(len(xyz), 16, 450) if mode == "deskew" else None
            rp = py.process(xyz, lab, ts, kitti=(mode == "kitti"), deskew=(mode == "deskew"))
            ro = orc.process(xyz, lab, ts, mode == "kitti")
            for a in ("n_in", "n_cluster", "n_ds", "n_src", "map_voxels", "iterations"):
                assert rp[a] == ro[a], f"{mode} scan {k}: {a} {rp[a]} != {ro[a]}"
            D = np.linalg.inv(pose_matrix(ro["pose"])) @ rp["pose"]
            assert np.linalg.norm(D[:3, 3]) < 1e-8 and Rotation.from_matrix(D[:3, :3]).magnitude() < 1e-9, (mode, k)
