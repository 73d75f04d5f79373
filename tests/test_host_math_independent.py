"""Independent checks of the host-side maths of the library (no GPU needed: these entry points take no context).

The product's SolveSVD / RansacAlignment (sg-slam_amd/csrc/graph_map.h) and the oracle's (oracle/sgo_core.hpp) were
written by the same hand, so "product == oracle" proves little there.  Here both are compared with restatements that
share no code with either: numpy's LAPACK SVD for the Kabsch solve of core/CommonUtils.cpp:264-316, and a pure-Python
Mersenne twister + the libstdc++ uniform_int_distribution rule for the RANSAC of core/CommonUtils.cpp:115-204."""
import numpy as np
import pytest

import oracle_py as O
from sg_slam_amd import api


def kabsch_numpy(p1, p2):
    """core/CommonUtils.cpp:264-316 with numpy: W = sum (p1 - m1)(p2 - m2)^T = U S V^T, R = V E U^T, t = m2 - R m1."""
    p1, p2 = np.asarray(p1, float), np.asarray(p2, float)
    m1, m2 = p1.mean(axis=0), p2.mean(axis=0)
    W = (p1 - m1).T @ (p2 - m2)
    U, S, Vt = np.linalg.svd(W)
    V = Vt.T
    E = np.eye(3)
    if np.linalg.det(V @ U.T) < 0:
        E[2, 2] = -1
    R = V @ E @ U.T
    return R, m2 - R @ m1


class MT19937:
    """std::mt19937 (32-bit Mersenne twister, init_genrand seeding), written from the published recurrence."""

    def __init__(self, seed):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def __call__(self):
        if self.idx >= 624:
            mt = self.mt
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def uniform_int(gen, n):
    """std::uniform_int_distribution<int>(0, n - 1) over a 32-bit generator as libstdc++ 11 draws it
    (bits/uniform_int_dist.h: Lemire's nearly-divisionless rule, _S_nd)."""
    prod = gen() * n
    low = prod & 0xFFFFFFFF
    if low < n:
        threshold = ((1 << 32) - n) % n
        while low < threshold:
            prod = gen() * n
            low = prod & 0xFFFFFFFF
    return prod >> 32


def ransac_python(m1, m2, nodes1, nodes2, max_iter, seed):
    """core/CommonUtils.cpp:115-204: 500 triplets, Kabsch per triplet (numpy), a node of graph 1 is an inlier when
    one of the 27 voxels (0.2 m held in a FLOAT, :126) around its image holds a node of graph 2."""
    gen = MT19937(seed)
    vs = np.float64(np.float32(0.2))  # `point / voxel_size` promotes the float to double
    key = lambda p: tuple(int(v) for v in np.trunc(np.asarray(p) / vs))
    occupied = {key(p) for p in nodes2}
    best, best_T = 0, (np.eye(3), np.zeros(3))
    for _ in range(max_iter):
        i1, i2, i3 = uniform_int(gen, len(m1)), uniform_int(gen, len(m1)), uniform_int(gen, len(m1))
        if len({i1, i2, i3}) < 2:  # all three equal: W = 0, Eigen's U = V = I
            R, t = np.eye(3), m2[i1] - m1[i1]
        else:
            R, t = kabsch_numpy(m1[[i1, i2, i3]], m2[[i1, i2, i3]])
        inl = 0
        for p in nodes1:
            kx, ky, kz = key(R @ p + t)
            inl += any((kx + a, ky + b, kz + c) in occupied for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1))
        if inl > best:
            best, best_T = inl, (R, t)
    return best_T[0], best_T[1], float(np.float32(best) / np.float32(len(nodes1))), best


def test_mt19937_known_answer():
    g = MT19937(5489)  # the C++ standard's check value: the 10000th output of a default-seeded mt19937
    for _ in range(9999):
        g()
    assert g() == 4123659995


@pytest.mark.parametrize("impl", [api.solve_svd, O.solve_svd])
def test_solve_svd_against_numpy_kabsch(impl):
    rng = np.random.default_rng(11)
    for trial in range(40):
        n = int(rng.integers(3, 12))
        T = O.se3_exp(rng.normal(size=6) * [2, 2, 2, 0.8, 0.8, 0.8])
        P = rng.normal(size=(n, 3)) * 10
        Q = O.se3_act(T, P) + rng.normal(size=(n, 3)) * 0.05
        R, t = impl(P, Q)
        Rn, tn = kabsch_numpy(P, Q)
        assert np.abs(R - Rn).max() < 1e-10 and np.abs(t - tn).max() < 1e-9, trial
        assert abs(np.linalg.det(R) - 1) < 1e-12 and np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    # three points (rank 2: the centred points are coplanar) with a mirrored target: the determinant rule decides
    P = np.array([[1.0, 0, 0], [0, 1, 0], [-1, -1, 0.3]])
    Q = P * [1, -1, 1]
    R, t = impl(P, Q)
    Rn, tn = kabsch_numpy(P, Q)
    assert np.abs(R - Rn).max() < 1e-10 and np.abs(t - tn).max() < 1e-10 and abs(np.linalg.det(R) - 1) < 1e-12


@pytest.mark.parametrize("impl", [api.ransac_alignment, O.ransac_alignment])
def test_ransac_against_python_restatement(impl):
    rng = np.random.default_rng(5)
    T = O.se3_exp([0.5, -1.0, 0.3, 0.1, -0.2, 0.4])
    nodes1 = rng.uniform(-40, 40, size=(30, 3))
    nodes2 = O.se3_act(T, nodes1)
    idx = rng.permutation(30)[:12]
    m1, m2 = nodes1[idx].copy(), nodes2[idx].copy()
    m2[:4] += rng.normal(size=(4, 3)) * 3  # a third of the matches is wrong
    for seed in (1, 7, 20251031):
        R, t, score, best = impl(m1, m2, nodes1, nodes2, 500, seed)
        Rp, tp, sp, bp = ransac_python(m1, m2, nodes1, nodes2, 500, seed)
        assert best == bp and score == sp, (seed, best, bp)
        assert np.abs(R - Rp).max() < 1e-9 and np.abs(t - tp).max() < 1e-8
    # few matches, many repeated draws (degenerate triplets included)
    R, t, score, best = impl(m1[4:7], m2[4:7], nodes1, nodes2, 60, 3)
    Rp, tp, sp, bp = ransac_python(m1[4:7], m2[4:7], nodes1, nodes2, 60, 3)
    assert best == bp and score == sp and np.abs(R - Rp).max() < 1e-9 and np.abs(t - tp).max() < 1e-8
