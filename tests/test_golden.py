"""Committed fixtures (tests/golden, made by tests/golden/make_golden.py).  PARITY UNPINNED: the
reference has no vectors of its own for this path; known_answers.npz follows from its formulas by
hand, mini_sequence.npz is the oracle's output on a seeded sequence (regression pin for the oracle
and an independent target for the HIP path)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_py as O  # noqa: E402
from sg_slam_amd import synth  # noqa: E402
from helpers import pose_err  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _known():
    return np.load(os.path.join(GOLD, "known_answers.npz"), allow_pickle=False)


def _mini():
    return np.load(os.path.join(GOLD, "mini_sequence.npz"), allow_pickle=False)


def test_oracle_known_answers():
    g = _known()
    assert np.array_equal(np.array([O.voxel_hash(k) for k in g["hash_keys"]], dtype=np.uint32), g["hash_values"])
    assert np.array_equal(O.voxel_keys(g["key_points"], float(g["key_voxel"])), g["key_values"])


def test_oracle_reproduces_mini_sequence():
    g = _mini()
    slam = O.Slam(O.kitti_config())
    for k in range(int(g["frames"])):
        xyzi, raw, lab = synth.scan(k, "mini")
        r = slam.process(xyzi[:, :3].astype(np.float64), lab)
        got = [r["n_in"], r["n_cluster"], r["n_ds"], r["n_src"], r["iterations"], r["map_voxels"], r["n_boxes"]]
        assert got == g["counts"][k].tolist(), k
        et, er = pose_err(g["poses"][k], r["pose"])
        assert et < 1e-9 and er < 1e-10, (k, et, er)  # same code, different OpenMP thread counts
    assert np.array_equal(slam.boxes(), g["last_boxes"])
    for which in (1, 2, 3):
        x, l = slam.cloud(which)
        assert np.array_equal(x, g[f"cloud{which}_xyz"]) and np.array_equal(l, g[f"cloud{which}_label"])


@pytest.mark.gpu
def test_hip_known_answers():
    from conftest import make_ctx
    g = _known()
    ctx = make_ctx()
    assert np.array_equal(ctx.voxel_keys(g["key_points"], float(g["key_voxel"])), g["key_values"])
    ctx.close()


@pytest.mark.gpu
def test_hip_reproduces_mini_sequence():
    from conftest import make_ctx
    g = _mini()
    ctx = make_ctx()
    for k in range(int(g["frames"])):
        xyzi, raw, lab = synth.scan(k, "mini")
        r = ctx.process_scan_f32(xyzi, lab, None, "")
        got = [r["n_in"], r["n_cluster"], r["n_ds"], r["n_src"], r["map_voxels"], r["n_boxes"]]
        assert got == g["counts"][k][[0, 1, 2, 3, 5, 6]].tolist(), k
        et, er = pose_err(g["poses"][k], r["pose"])
        assert et < 1e-4 and er < 1e-5, (k, et, er)  # BASELINE.json north_star tolerance
    assert np.array_equal(ctx.boxes(), g["last_boxes"])
    for which in (1, 2, 3):  # voxel-downsampled sets: bit-exact incl. order
        x, l = ctx.cloud(which)
        assert np.array_equal(x, g[f"cloud{which}_xyz"]) and np.array_equal(l, g[f"cloud{which}_label"])
    ctx.close()


def _next():
    return np.load(os.path.join(GOLD, "next_rows.npz"), allow_pickle=False)


def _check_next_rows(scan_descriptor, fast_icp, solve_svd, ransac):
    g = _next()
    assert np.array_equal(scan_descriptor(), g["scan_descriptor"])
    pose, it = fast_icp(g["ficp_a"], g["ficp_b"], g["ficp_init"])
    et, er = pose_err(g["ficp_pose"], pose)
    assert et < 1e-6 and er < 1e-7 and abs(it - int(g["ficp_iterations"])) <= 1
    R, t = solve_svd(g["match1"][3:], g["match2"][3:])
    assert np.abs(R - g["svd_R"]).max() < 1e-12 and np.abs(t - g["svd_t"]).max() < 1e-11
    Tt = g["svd_truth"]  # the pairs were made with this transform: R, t must be it
    assert np.abs(R @ g["match1"][3:].T + t[:, None] - O.se3_act(Tt, g["match1"][3:]).T).max() < 1e-11
    R, t, score, best = ransac(g["match1"], g["match2"], g["nodes1"], g["nodes2"])
    assert best == int(g["ransac_best"]) == 24 and score == float(g["ransac_score"])
    assert np.abs(R - g["ransac_R"]).max() < 1e-12 and np.abs(t - g["ransac_t"]).max() < 1e-10


def test_oracle_reproduces_next_rows():
    m = _mini()
    slam = O.Slam(O.kitti_config())
    for k in range(int(m["frames"])):
        xyzi, raw, lab = synth.scan(k, "mini")
        slam.process(xyzi[:, :3].astype(np.float64), lab)

    def fast_icp(a, b, init):
        p, info = O.fast_icp(a, b, init, 1.0, 5, 0.5)
        return p, info["iterations"]
    _check_next_rows(lambda: O.scan_descriptor(slam.boxes(), *slam.cloud(1)), fast_icp, O.solve_svd,
                     lambda a, b, c, d: O.ransac_alignment(a, b, c, d, 500, 7))


@pytest.mark.gpu
def test_hip_reproduces_next_rows():
    from conftest import make_ctx
    from sg_slam_amd import api
    m = _mini()
    ctx = make_ctx()
    for k in range(int(m["frames"])):
        xyzi, raw, lab = synth.scan(k, "mini")
        ctx.process_scan_f32(xyzi, lab, None, "")
    desc = ctx.scan_descriptor()  # before anything else touches the context's last scan
    _check_next_rows(lambda: desc, lambda a, b, init: ctx.FastIcpGetTrans(a, b, init, 1.0, 5, 0.5), api.solve_svd,
                     lambda a, b, c, d: api.ransac_alignment(a, b, c, d, 500, 7))
    ctx.close()
