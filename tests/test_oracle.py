"""Pins the CPU oracle (oracle/) with known-answer tests built from the reference's source semantics
(SURVEY.md 2.4 / section 4).  The reference ships no tests or golden vectors, so every expected
value here is derived independently (python big-int arithmetic, numpy/scipy, hand-built cases);
`parity unpinned` against a compiled reference (it cannot be built in this environment)."""
import math

import numpy as np
import pytest
from scipy.linalg import expm
from scipy.spatial.transform import Rotation as R

import oracle_py as O
from helpers import cluster_reference, pose_matrix, ref_hash, robin_order, trunc_key

RNG = np.random.default_rng(20251031)


# ---------------- third-party maths restated (Sophus / Eigen) ----------------
def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def test_se3_exp_matches_matrix_exponential():
    for scale in (1e-12, 1e-6, 0.3, 2.5):
        for _ in range(10):
            t6 = RNG.normal(size=6) * scale
            T = pose_matrix(O.se3_exp(t6))
            X = np.zeros((4, 4))
            X[:3, :3] = hat(t6[3:])
            X[:3, 3] = t6[:3]
            assert np.allclose(T, expm(X), atol=1e-12)


def test_se3_log_inverts_exp_and_handles_small_angles():
    for scale in (0.0, 1e-11, 1e-5, 0.5, 3.0):
        t6 = RNG.normal(size=6) * scale
        nw = np.linalg.norm(t6[3:])
        if nw > 3.0:  # log returns the representative with |omega| <= pi
            t6[3:] *= 3.0 / nw
        assert np.allclose(O.se3_log(O.se3_exp(t6)), t6, atol=1e-10)


def test_se3_group_ops():
    a, b = O.se3_exp(RNG.normal(size=6)), O.se3_exp(RNG.normal(size=6))
    assert np.allclose(pose_matrix(O.se3_mul(a, b)), pose_matrix(a) @ pose_matrix(b), atol=1e-12)
    assert np.allclose(pose_matrix(O.se3_inv(a)), np.linalg.inv(pose_matrix(a)), atol=1e-12)
    p = RNG.normal(size=(5, 3)) * 50
    assert np.allclose(O.se3_act(a, p), p @ pose_matrix(a)[:3, :3].T + pose_matrix(a)[:3, 3], atol=1e-12)


def test_ldlt_solve_and_zero_system():
    M = RNG.normal(size=(6, 6))
    A = M @ M.T + 0.1 * np.eye(6)
    b = RNG.normal(size=6)
    assert np.allclose(O.ldlt6_solve(A, b), np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # Eigen's LDLT treats an all-zero matrix as rank 0: x = 0 (no correspondences => identity step)
    assert np.array_equal(O.ldlt6_solve(np.zeros((6, 6)), b), np.zeros(6))
    # indefinite / pivoting case
    A2 = np.diag([1e-3, 5.0, -2.0, 1.0, 7.0, 0.5]) + 0.01 * (M + M.T)
    assert np.allclose(O.ldlt6_solve(A2, b), np.linalg.solve(A2, b), rtol=1e-8, atol=1e-10)


def test_sym_eig_matches_numpy():
    for n in (1, 2, 5, 17, 40):
        M = (RNG.random((n, n)) < 0.3).astype(float)
        A = np.maximum(M, M.T)
        np.fill_diagonal(A, 1.0)
        w, v = O.sym_eig(A)
        assert np.allclose(w, np.linalg.eigvalsh(A), atol=1e-9)
        assert np.allclose(A @ v, v * w, atol=1e-8)
        assert np.allclose(v.T @ v, np.eye(n), atol=1e-9)


# ---------------- voxel keys / hash ----------------
def test_voxel_keys_truncate_toward_zero():
    pts = np.array([[0.19, -0.19, 0.0], [-0.0, 0.2, -0.2], [0.6, -0.6, 0.3], [-1.0, 1.0, 0.9999999],
                    [99.99, -99.99, 1e-300], [0.30000000000000004, 0.3, -0.30000000000000004]])
    k = O.voxel_keys(pts, 0.2)
    # cells straddling 0 are double width: +-0.19 -> 0 ;  x/0.2 stays a correctly rounded fp64 divide
    assert k[0].tolist() == [0, 0, 0]
    assert k[1].tolist() == [0, 1, -1]
    assert k[2].tolist() == [int(0.6 / 0.2), int(-0.6 / 0.2), int(0.3 / 0.2)]
    assert np.array_equal(k, trunc_key(pts, 0.2))
    big = RNG.normal(size=(5000, 3)) * 60
    for v in (0.2, 0.5, 1.0, 1.5):
        assert np.array_equal(O.voxel_keys(big, v), trunc_key(big, v))


def test_voxel_hash_known_answers():
    for k in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, -1, -1), (123, -456, 789), (-500, 499, -17), (2 ** 20, -2 ** 20, 5)]:
        assert O.voxel_hash(k) == ref_hash(k)
    assert ref_hash((1, 0, 0)) == 73856093 & ((1 << 20) - 1)
    assert ref_hash((-1, 0, 0)) == ((-73856093) & 0xFFFFFFFF) & ((1 << 20) - 1)


# ---------------- preprocessing ----------------
def test_preprocess_strict_bounds_and_label_filter():
    xyz = np.array([[5.0, 0, 0], [5.0000001, 0, 0], [100.0, 0, 0], [99.999, 0, 0], [30, 40, 0], [10, 0, 0], [10, 0, 0]])
    lab = np.array([1, 1, 1, 1, 19, 20, 25])
    ox, ol = O.preprocess(xyz, lab, 100.0, 5.0)
    assert ox.tolist() == [[5.0000001, 0, 0], [99.999, 0, 0], [30, 40, 0]]
    assert ol.tolist() == [1, 1, 19]
    ox, ol = O.preprocess(np.zeros((0, 3)), np.zeros(0, dtype=np.int32), 100.0, 5.0)
    assert len(ox) == 0


def test_correct_kitti_scan_is_a_rotation_about_p_cross_z():
    pts = RNG.normal(size=(200, 3)) * 20
    out = O.correct_kitti(pts)
    ang = 0.205 * math.pi / 180
    for p, q in zip(pts[:50], out[:50]):
        ax = np.cross(p, [0, 0, 1.0])
        ax /= np.linalg.norm(ax)
        assert np.allclose(q, R.from_rotvec(ax * ang).apply(p), atol=1e-12)
    # p parallel to z: normalized() leaves the zero vector, R degenerates to cos(a)*I (no NaN)
    z = O.correct_kitti(np.array([[0.0, 0.0, 7.0]]))
    assert np.allclose(z, [[0, 0, 7.0 * math.cos(ang)]], atol=1e-15)


def test_deskew_constant_velocity():
    pts = RNG.normal(size=(100, 3)) * 30
    ts = RNG.random(100)
    a = O.se3_exp([0, 0, 0, 0, 0, 0.3])
    b = O.se3_mul(a, O.se3_exp([1.0, 0.1, 0, 0, 0.01, 0.02]))
    out = O.deskew(pts, ts, a, b)
    delta = O.se3_log(O.se3_mul(O.se3_inv(a), b))
    for i in range(0, 100, 9):
        assert np.allclose(out[i], O.se3_act(O.se3_exp((ts[i] - 0.5) * delta), pts[i:i + 1])[0], atol=1e-12)
    # mid-scan points do not move
    mid = O.deskew(pts[:3], np.full(3, 0.5), a, b)
    assert np.allclose(mid, pts[:3], atol=1e-13)


# ---------------- voxel downsampling ----------------
def test_downsample_first_point_wins_and_robin_order():
    for n, spread in ((1, 1.0), (2, 0.1), (500, 3.0), (3000, 10.0), (4096, 2.0)):
        xyz = RNG.normal(size=(n, 3)) * spread
        lab = RNG.integers(0, 20, n)
        for v in (0.2, 0.5, 1.5):
            ox, ol, oi, info = O.voxel_downsample(xyz, lab, v)
            keys = trunc_key(xyz, v)
            exp = robin_order(keys)
            assert oi.tolist() == exp
            assert np.array_equal(ox, xyz[exp]) and np.array_equal(ol, lab[exp])
            # first point of every voxel, each voxel exactly once
            first = {}
            for i, k in enumerate(map(tuple, keys)):
                first.setdefault(k, i)
            assert sorted(exp) == sorted(first.values())
            assert info["grow_by_probe"] == 0


def _keys_with_home(B, homes, per_home, voxel=1.0):
    """points (voxel centres) whose reference hash & (B-1) hits the requested home buckets."""
    out = []
    need = {h: per_home for h in homes}
    k = 0
    while any(need.values()):
        key = (k % 97 - 48, (k // 97) % 89 - 44, k // (97 * 89) - 3)
        h = ref_hash(key) & (B - 1)
        if need.get(h, 0) > 0:
            need[h] -= 1
            out.append([(c + 0.5) * voxel if c >= 0 else (c - 0.5) * voxel for c in key])
        k += 1
    return np.array(out)


def test_downsample_order_with_engineered_ties_and_wraparound():
    # 40 points -> B = 128 buckets.  Clusters of same-home voxels around the table end force both the
    # "displaced element hops over equal-distance residents" reordering and the wrap to bucket 0.
    B = 128
    pts = _keys_with_home(B, [125, 126, 127, 0, 1, 60, 61], 5)
    assert len(pts) == 35
    filler = RNG.normal(size=(5, 3)) * 30
    for trial in range(20):
        xyz = np.concatenate([pts, filler])
        xyz = xyz[RNG.permutation(len(xyz))]
        lab = np.zeros(len(xyz), dtype=np.int32)
        ox, ol, oi, info = O.voxel_downsample(xyz, lab, 1.0)
        assert info["buckets"] == B
        assert oi.tolist() == robin_order(trunc_key(xyz, 1.0))
        assert info["wrapped"] > 0


def test_downsample_canonical_mode_is_sorted_first_indices():
    xyz = RNG.normal(size=(2000, 3)) * 5
    lab = np.zeros(2000, dtype=np.int32)
    _, _, oi, _ = O.voxel_downsample(xyz, lab, 0.5, order_mode=1)
    _, _, oi0, _ = O.voxel_downsample(xyz, lab, 0.5, order_mode=0)
    assert oi.tolist() == sorted(oi0.tolist())


# ---------------- voxel map ----------------
def test_map_cap_and_insertion_order():
    m = O.Map(1.0, 100.0, 20)
    assert m.empty()
    pts = np.zeros((50, 4))
    pts[:, 0] = 0.5 + np.arange(50) * 0.001  # all in voxel (0,0,0)
    pts[:, 3] = np.arange(50) % 20
    m.add_points(pts[:7])
    m.add_points(pts[7:])
    keys, cnt, p = m.dump()
    assert keys.tolist() == [[0, 0, 0]] and cnt.tolist() == [20]
    assert np.array_equal(p[0], pts[:20])  # first come, in order, capped at 20
    assert not m.empty() and m.size() == (1, 20)
    m.clear()
    assert m.empty()


def test_map_prune_uses_oldest_point_and_strict_greater():
    m = O.Map(1.0, 10.0, 20)
    m.add_points(np.array([[9.5, 0.5, 0.5, 1], [9.9, 0.5, 0.5, 2]]))      # voxel (9,0,0), oldest at 9.5
    m.add_points(np.array([[10.0, 0.2, 0.0, 3]]))                          # voxel (10,0,0), exactly at max
    m.add_points(np.array([[10.4, 0.2, 0.0, 4], [10.01, 0.2, 0.0, 5]]))   # same voxel, newer points
    m.add_points(np.array([[-10.5, 0.0, 0.0, 6]]))                        # beyond
    m.prune([0.0, 0.0, 0.0])
    keys, cnt, _ = m.dump()
    # (10,0,0): oldest point is at distance^2 = 100.04 > 100 -> removed; (-10,0,0) removed; (9,0,0) stays
    assert keys.tolist() == [[9, 0, 0]]
    m2 = O.Map(1.0, 10.0, 20)
    m2.add_points(np.array([[10.0, 0.0, 0.0, 3], [10.9, 0.0, 0.0, 3]]))
    m2.prune([0.0, 0.0, 0.0])
    assert m2.size()[0] == 1  # distance == max_distance is kept (strict '>')


def test_prune_quirk_mode_only_differs_beyond_range():
    # mode 1 replays the reference's erase-while-iterating (an element shifted into the current bucket
    # is skipped for one sweep); mode 0 is the pure predicate the GPU implements.
    pts = np.concatenate([RNG.normal(size=(4000, 3)) * 40, np.zeros((4000, 1))], axis=1)
    a, b = O.Map(1.0, 50.0, 20, prune_mode=0), O.Map(1.0, 50.0, 20, prune_mode=1)
    for m in (a, b):
        m.add_points(pts)
        m.prune([0.0, 0.0, 0.0])
    ka, _, pa = a.dump()
    kb, _, pb = b.dump()
    sa = set(map(tuple, ka.tolist()))
    sb = set(map(tuple, kb.tolist()))
    assert sa <= sb
    for k in sb - sa:  # survivors of the quirk are all voxels the predicate would have removed
        first = pb[[i for i, kk in enumerate(kb.tolist()) if tuple(kk) == k][0], 0, :3]
        assert np.dot(first, first) > 50.0 ** 2


def test_prune_quirk_does_not_reach_the_poses_over_a_drive():
    """Row a12 quantified on the CPU: the whole pipeline under the reference's erase-while-iterating rule
    (prune_mode=1, F/VoxelHashMap.cpp:178-186) and under the pure predicate (prune_mode=0, what the library does), 120
    full-size scans.  Voxels do survive an extra scan under the reference rule, but they lie beyond max_range of the
    pose that left them: no correspondence count and no pose changes, bit for bit.  (220 / 1 000 scans: 39 / 169
    surviving voxel-scans in 31 / 144 scans, same verdict - DESIGN.md section 2; the GPU suite runs the library against
    both modes.)"""
    from sg_slam_amd import synth
    a, b = O.Slam(O.kitti_config(prune_mode=0, enable_graph=0)), O.Slam(O.kitti_config(prune_mode=1, enable_graph=0))
    survivors = 0
    for k in range(120):
        xyzi, _, lab = synth.scan(k, "hdl64")
        xyz = xyzi[:, :3].astype(np.float64)
        ra, rb = a.process(xyz, lab), b.process(xyz, lab)
        assert np.array_equal(ra["pose"], rb["pose"]) and ra["iterations"] == rb["iterations"], k
        assert ra["correspondences"] == rb["correspondences"], k
        assert rb["map_voxels"] >= ra["map_voxels"]
        survivors += rb["map_voxels"] - ra["map_voxels"]
    assert survivors > 0  # the quirk does fire on this drive; it just never matters


def test_correspondences_27_neighbourhood_and_first_wins_tie():
    m = O.Map(1.0, 100.0, 20)
    # two map points equidistant from the query: the one visited first (x-outer, z-inner voxel order,
    # then insertion order) must win thanks to the strict '<'
    m.add_points(np.array([[1.5, 0.5, 0.5, 7], [-0.5, 0.5, 0.5, 8]]))
    src, tgt, qi, cand = m.get_correspondences(np.array([[0.5, 0.5, 0.5, 0]]), 10.0)
    assert cand == 2 and tgt[0].tolist() == [-0.5, 0.5, 0.5, 8]  # voxel (-1,0,0) is visited before (1,0,0)
    # same voxel: insertion order decides
    m2 = O.Map(1.0, 100.0, 20)
    m2.add_points(np.array([[0.75, 0.5, 0.5, 1], [0.25, 0.5, 0.5, 2]]))
    _, tgt, _, _ = m2.get_correspondences(np.array([[0.5, 0.5, 0.5, 0]]), 10.0)
    assert tgt[0, 3] == 1
    # points two voxels away are invisible; the distance gate is strict
    m3 = O.Map(1.0, 100.0, 20)
    m3.add_points(np.array([[2.5, 0.5, 0.5, 1]]))
    assert len(m3.get_correspondences(np.array([[0.5, 0.5, 0.5, 0]]), 10.0)[0]) == 0
    m3.add_points(np.array([[1.5, 0.5, 0.5, 1]]))
    assert len(m3.get_correspondences(np.array([[0.5, 0.5, 0.5, 0]]), 1.0)[0]) == 0
    assert len(m3.get_correspondences(np.array([[0.5, 0.5, 0.5, 0]]), 1.0000001)[0]) == 1
    # negative coordinates use truncation: -0.5 lives in voxel 0 together with +0.5
    m4 = O.Map(1.0, 100.0, 20)
    m4.add_points(np.array([[-0.5, -0.5, -0.5, 1], [0.5, 0.5, 0.5, 2]]))
    assert m4.size() == (1, 2)


# ---------------- registration ----------------
def _numpy_linear_system(src, tgt, th):
    JTJ = np.zeros((6, 6))
    JTr = np.zeros(6)
    for s, t in zip(src, tgt):
        r = s[:3] - t[:3]
        J = np.hstack([np.eye(3), -hat(s[:3])])
        w = th * th / (th + r @ r) ** 2
        lab = int(s[3])
        ws = 1.2 if lab in (16, 18, 19) else (1.0 if 0 <= lab < 20 else 0.0)
        JTJ += J.T @ J * w
        JTr += J.T @ r * w * ws
    return JTJ, JTr


def test_linear_system_semantic_weight_only_on_jtr():
    m = O.Map(1.0, 100.0, 20)
    mp = np.concatenate([RNG.random((400, 3)) * 6, np.zeros((400, 1))], axis=1)
    m.add_points(mp)
    q = np.concatenate([RNG.random((60, 3)) * 6, RNG.integers(0, 26, (60, 1)).astype(float)], axis=1)
    src, tgt, qi, _ = m.get_correspondences(q, 3.0)
    out, n = m.build_linear_system(q, 3.0, 0.5)
    JTJ, JTr = _numpy_linear_system(src, tgt, 0.5)
    assert n == len(src)
    assert np.allclose(out[:21], JTJ[np.triu_indices(6)], rtol=1e-10)
    assert np.allclose(out[21:], JTr, rtol=1e-9, atol=1e-12)
    # relabelling every source point to a weight-0 class zeroes JTr but leaves JTJ untouched
    q0 = q.copy()
    q0[:, 3] = 22
    out0, _ = m.build_linear_system(q0, 3.0, 0.5)
    assert np.allclose(out0[:21], out[:21], rtol=1e-12) and np.all(out0[21:] == 0)


def test_icp_recovers_known_transform_and_empty_map_returns_guess():
    scene = np.concatenate([RNG.random((6000, 3)) * [40, 40, 6], np.zeros((6000, 1))], axis=1)
    m = O.Map(1.0, 100.0, 20)
    m.add_points(scene)
    T_true = O.se3_exp([0.3, -0.2, 0.05, 0.01, -0.02, 0.03])
    src = scene[::5].copy()
    src[:, :3] = O.se3_act(O.se3_inv(T_true), src[:, :3])
    pose, info = m.register_frame(src, np.array([0, 0, 0, 1.0, 0, 0, 0]), 3.0, 0.3)
    assert info["iterations"] >= 2
    assert np.allclose(pose_matrix(pose), pose_matrix(T_true), atol=1e-6)
    empty = O.Map(1.0, 100.0, 20)
    g = O.se3_exp([1, 2, 3, 0.1, 0.2, 0.3])
    pose, info = empty.register_frame(src, g, 3.0, 0.3)
    assert np.array_equal(pose, g) and info["iterations"] == 0


def test_adaptive_threshold_sequence():
    # first scans use initial_threshold until the platform moved > 5*min_motion_th (P/SemGraphSLAM.cpp:147-159)
    from sg_slam_amd import synth
    s = O.Slam(O.kitti_config(enable_graph=0))
    sig = []
    for k in range(6):
        xyzi, raw, lab = synth.scan(k, "mini")
        sig.append(s.process(xyzi[:, :3].astype(np.float64), lab)["sigma"])
    assert sig[0] == 2.0 and sig[1] == 2.0
    assert all(x > 0 for x in sig)


# ---------------- clustering / graph ----------------
def test_cluster_labels_match_the_literal_algorithm():
    # two blobs + scattered points of one class, checked against the plain-python Appendix-C restatement
    a = RNG.normal(size=(150, 3)) * [0.8, 0.4, 0.4] + [12, 3, -0.8]
    b = RNG.normal(size=(120, 3)) * [0.5, 0.9, 0.4] + [-9, -14, -0.6]
    c = RNG.normal(size=(60, 3)) * [6, 6, 0.5] + [20, -5, -1.0]
    xyz = np.concatenate([a, b, c])
    xyz = xyz[RNG.permutation(len(xyz))]
    lab = np.full(len(xyz), 1, dtype=np.int32)
    boxes, pc, nfg, nbg = O.cluster_points(xyz, lab, 2.0, 0.5, 2.0)
    assert nfg == len(xyz) and nbg == 0
    assert pc.tolist() == cluster_reference(xyz.tolist(), 2.0, 0.5, 2.0)
    assert len(boxes) >= 2 and set(boxes[:, 6]) == {1.0}
    for bx in boxes:
        assert bx[7] >= 10 and bx[3] >= 0.5 and bx[4] >= 0.5 and bx[5] >= 0.3


def test_cluster_class_routing_and_filters():
    blob = lambda c, n: RNG.normal(size=(n, 3)) * 0.3 + c
    xyz = np.concatenate([blob([10, 0, -1], 40), blob([0, 12, 0], 40), blob([-8, -8, 1], 40), blob([15, 15, -1], 5),
                          blob([30, 0, 0], 30)])
    lab = np.concatenate([np.full(40, 16), np.full(40, 18), np.full(40, 4), np.full(5, 1), np.full(30, 13)]).astype(np.int32)
    boxes, pc, nfg, nbg = O.cluster_points(xyz, lab, 2.0, 0.5, 2.0)
    assert nfg == 125 and nbg == 30
    labels = sorted(boxes[:, 6].tolist())
    # trunk -> 2, pole -> 3, truck -> 1 (vehicle); the 5-point car cluster is dropped (< 10 points)
    assert labels == [1.0, 2.0, 3.0]
    # classes are emitted in class order 1,4,5,16,18
    assert boxes[:, 6].tolist() == [1.0, 2.0, 3.0]


def test_graph_histogram_bins_and_edges():
    # nodes on a line: distances 0, 3, 7, 61 -> bins int(d/2): self edge in bin 0, 3 -> 1, 7 -> 3, 61 no edge
    boxes = np.array([[0, 0, 0, 1, 1, 1, 1, 20], [3, 0, 0, 1, 1, 1, 2, 20], [7, 0, 0, 1, 1, 1, 3, 20], [61, 0, 0, 1, 1, 1, 1, 20]], dtype=float)
    g = O.build_graph(boxes, 60.0, 30, 30, 20.0)
    assert g["edges"].tolist() == [[0, 1], [0, 2], [1, 2], [1, 3], [2, 3]]
    assert np.allclose(g["edge_weights"][:2], [(60 - 3) / 60, (60 - 7) / 60])
    d0 = g["desc"][0, :90]
    exp = np.zeros(90)
    exp[0] = 1       # vehicle block, self edge (distance 0) in bin 0
    exp[30 + 1] = 1  # trunk block, bin int(3/2)
    exp[60 + 3] = 1  # pole block, bin int(7/2)
    assert np.allclose(d0, exp / np.linalg.norm(exp), atol=1e-7)
    assert g["desc"].shape == (4, 120)
    assert g["counts"].tolist() == [2, 1, 1]
    # spectral half: rows sum to 1, zero padded beyond N columns
    assert np.allclose(g["desc"][:, 90:].sum(axis=1), 1.0, atol=1e-6)
    assert np.all(g["desc"][:, 94:] == 0)
    # sub-triangles: node 0 has neighbours {1,2} within 20 m -> needs > 2 neighbours, so none
    assert g["tri_counts"].tolist() == [0, 0, 0, 0]
    # empty input
    g0 = O.build_graph(np.zeros((0, 8)))
    assert g0["desc"].shape[0] == 0


def test_graph_sub_triangles_sorted_float_sides():
    c = np.array([[0, 0, 0], [3, 0, 0], [0, 4, 0], [5, 5, 0]], dtype=float)
    boxes = np.concatenate([c, np.ones((4, 3)), np.ones((4, 1)), np.full((4, 1), 20.0)], axis=1)
    g = O.build_graph(boxes, 60.0, 30, 30, 20.0)
    assert g["tri_counts"].tolist() == [3, 3, 3, 3]
    t0 = g["triangles"][:3]
    exp = []
    for m, n in ((1, 2), (1, 3), (2, 3)):
        s = sorted(np.float32(x) for x in (np.linalg.norm(c[0] - c[m]), np.linalg.norm(c[0] - c[n]), np.linalg.norm(c[m] - c[n])))
        exp.append([float(x) for x in s])
    assert np.allclose(t0, exp, atol=0)


def test_fast_icp_recovers_a_known_transform_and_uses_unit_weights():
    """loopclosure/FastIcp.cpp: B is a point lattice, A the same points seen from a displaced frame."""
    rng = np.random.default_rng(3)
    g = np.stack(np.meshgrid(np.arange(-10, 10, 1.3), np.arange(-10, 10, 1.3), np.arange(-2, 3, 1.1), indexing="ij"), -1).reshape(-1, 3)
    B = np.concatenate([g + rng.normal(size=g.shape) * 0.05, rng.integers(0, 26, (len(g), 1))], axis=1)
    T = O.se3_exp([0.12, -0.08, 0.03, 0.004, -0.002, 0.01])
    A = B.copy()
    A[:, :3] = O.se3_act(O.se3_inv(T), B[:, :3])
    pose, info = O.fast_icp(A, B, O.se3_exp([0, 0, 0, 0, 0, 0]), 1.0, 5, 0.5)
    assert np.abs(pose - T).max() < 1e-9 and 2 <= info["iterations"] < 20
    assert info["correspondences"] >= len(B)  # every point pairs with itself in every iteration
    # labels play no role (no class weight): relabelling everything as an "ignored" class changes nothing
    A2, B2 = A.copy(), B.copy()
    A2[:, 3] = 25; B2[:, 3] = 0
    pose2, _ = O.fast_icp(A2, B2, O.se3_exp([0, 0, 0, 0, 0, 0]), 1.0, 5, 0.5)
    assert np.array_equal(pose, pose2)
    # the voxel size is held in a float: 0.8f != 0.8, a point at 4 * 0.8 lands in voxel 3 of the float grid
    assert int(3.2 / float(np.float32(0.8))) == 3 and int(3.2 / 0.8) == 4
    # cap 1 keeps only the first point of a voxel
    two = np.array([[0.1, 0.1, 0.1, 1], [0.6, 0.6, 0.6, 1]])
    q = np.array([[0.62, 0.6, 0.6, 1.0]])
    p_cap1, i1 = O.fast_icp(q, two, O.se3_exp([0] * 6), 1.0, 1, 2.0)
    p_cap5, i5 = O.fast_icp(q, two, O.se3_exp([0] * 6), 1.0, 5, 2.0)
    assert abs(p_cap5[4] + 0.02) < 1e-9 and abs(p_cap1[4] + 0.52) < 1e-9


def test_scan_descriptor_known_answers():
    """loopclosure/LoopClosure.cpp:15-188 restated: hand-checkable layout of the 231-D scan descriptor."""
    import math
    # fastAtan2 is a degree-valued polynomial approximation, accurate to ~0.01 deg, exact on the axes
    assert O.fast_atan2(0, 1) == 0.0 and O.fast_atan2(1, 0) == 90.0 and O.fast_atan2(0, -1) == 180.0 and O.fast_atan2(-1, 0) == 270.0
    for y, x in [(1, 1), (0.3, -2), (-5, 0.1), (-0.2, -0.9), (2, 7)]:
        assert abs(O.fast_atan2(y, x) - math.degrees(math.atan2(y, x)) % 360) < 0.02
    # two vehicles 3 m apart, a trunk 10 m from the first: bins = int(edge / 2.0)
    boxes = np.array([[0, 0, 0, 1, 1, 1, 1, 50], [3, 0, 0, 1, 1, 1, 1, 50], [0, 10, 0, 1, 1, 1, 2, 30]], dtype=np.float64)
    # background: building points in ring 0 at sectors 0 and 90, road point in ring 23
    pts = np.array([[5.5, 0, 0], [5.6, 0.001, 0], [0, 6.0, 0], [49.0, 0, 0], [60.0, 0, 0], [3.0, 0, 0]])
    lab = np.array([13, 13, 13, 9, 13, 9], dtype=np.int32)
    d = O.scan_descriptor(boxes, pts, lab).astype(np.float64)
    assert d.shape == (231,)
    g, b = d[:183], d[183:]
    nz = np.flatnonzero(g)
    # counts (vehicle, trunk, pole) ; vehicle-vehicle bin 1 (3 m) ; vehicle-trunk bins 5 (10 m) and 5 (sqrt(109) = 10.44 m)
    assert nz.tolist() == [0, 1, 3 + 1, 3 + 4 * 30 + 5]
    assert g[3 + 4 * 30 + 5] / g[3 + 1] == pytest.approx(2.0, rel=1e-6) and g[0] / g[1] == pytest.approx(2.0, rel=1e-6)
    assert np.flatnonzero(b).tolist() == [0, 24 + 23]          # building ring 0, road ring 23; 60 m and 3 m are out of range
    assert b[0] / b[47] == pytest.approx(2.0, rel=1e-6)        # two occupied sectors against one (5.5 and 5.6 share sector 0)
    assert np.linalg.norm(d) == pytest.approx(1.0, abs=1e-6)
    assert np.linalg.norm(g) == pytest.approx(np.linalg.norm(b), rel=1e-6)  # both halves are unit vectors before the final scaling


def test_graph_map_and_relocalization_known_answers():
    """frontend/GraphMap.cpp + core/CommonUtils.cpp restated: hand-checkable behaviour."""
    rng = np.random.default_rng(9)
    # SolveSVD recovers a rigid transform exactly from 3 and from many pairs, det = +1
    T = O.se3_exp([0.5, -1.0, 0.3, 0.1, -0.2, 0.4])
    for n in (3, 10):
        P = rng.normal(size=(n, 3)) * 5
        R, t = O.solve_svd(P, O.se3_act(T, P))
        assert np.abs(R @ P.T + t[:, None] - O.se3_act(T, P).T).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
    # RANSAC with a third of the matches wrong still finds the transform that explains every node
    nodes1 = rng.uniform(-40, 40, size=(30, 3)); nodes2 = O.se3_act(T, nodes1)
    idx = rng.permutation(30)[:12]
    m1, m2 = nodes1[idx].copy(), nodes2[idx].copy()
    m2[:4] += rng.normal(size=(4, 3)) * 3
    R, t, score, best = O.ransac_alignment(m1, m2, nodes1, nodes2, 500, 7)
    assert best == 30 and score == 1.0 and np.abs(R @ nodes1.T + t[:, None] - nodes2.T).max() < 1e-9
    assert O.ransac_alignment(m1, m2, nodes1, nodes2, 500, 7)[3] == best  # seeded: reproducible
    # a static scene of well separated instances: first scan adds them, second scan matches them all, none is added twice
    def box(x, y, label, pts=80):
        return [x, y, 0.5, 2.0, 2.0, 1.5, label, pts]
    scene = np.array([box(10 * i, 7 * (i % 3) - 5, 1 + i % 3) for i in range(12)], dtype=np.float64)
    gm = O.GraphMap(nn_vehicle=1.0, nn_pole=1.0, max_distance=150.0)
    ident = O.se3_exp([0] * 6)
    m, fresh = gm.track(scene, ident)
    assert not m[:, 0].any() and fresh.tolist() == list(range(12)) and len(gm.instances()) == 12
    m, fresh = gm.track(scene, ident)
    assert m[:, 0].all() and len(fresh) == 0 and len(gm.instances()) == 12
    assert np.array_equal(m[:, 1], m[:, 2])  # every node found itself
    # too small / too few points are not admitted (GraphMap.cpp:185-191); a second one of the same class nearby neither
    extra = np.array([box(200, 0, 1, pts=20), [210, 0, 0.5, 0.5, 2, 2, 1, 80], [220, 0, 0.5, 2, 2, 0.5, 2, 80], box(10.5, 2.2, 2)])
    m, fresh = gm.track(np.concatenate([scene, extra]), ident)
    assert len(gm.instances()) == 12 and len(fresh) == 0
    # moving 200 m away drops what is farther than max_distance from the sensor
    far = O.se3_exp([200.0, 0, 0, 0, 0, 0])
    gm.track(np.zeros((0, 8)), far)
    inst = gm.instances()
    assert 0 < len(inst) < 12 and np.linalg.norm(inst[:, 2:5] - [200, 0, 0], axis=1).max() <= 150.0 and inst[:, 2].min() >= 50.0
    # relocalization: the scene seen from a displaced pose gives back that pose
    gm = O.GraphMap(nn_vehicle=1.0, nn_pole=1.0, max_distance=200.0)
    gm.track(scene, ident)
    Tq = O.se3_exp([3.0, -2.0, 0.0, 0, 0, 0.3])
    seen = scene.copy()
    seen[:, :3] = O.se3_act(O.se3_inv(Tq), scene[:, :3])
    ok, pose, score, nm = gm.relocalize(seen, 0.4, 5)
    assert ok and nm >= 4 and score > 0.9
    assert np.abs(O.se3_act(pose, seen[:, :3]) - scene[:, :3]).max() < 1e-6
