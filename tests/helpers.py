"""Shared helpers for the parity tests (pure numpy / python; independent of the oracle's C++)."""
import numpy as np

M20 = (1 << 20) - 1


def ref_hash(k):
    """C/Coreutils.h:40-45 restated with Python ints (uint32 wraparound)."""
    x, y, z = (int(v) & 0xFFFFFFFF for v in k)
    return M20 & (((x * 73856093) & 0xFFFFFFFF) ^ ((y * 19349663) & 0xFFFFFFFF) ^ ((z * 83492791) & 0xFFFFFFFF))


def trunc_key(xyz, voxel):
    """trunc toward zero of the fp64 quotient (numpy division is IEEE, astype truncates)."""
    return np.trunc(np.asarray(xyz, dtype=np.float64) / np.float64(voxel)).astype(np.int64)


def pow2ceil(v):
    p = 1
    while p < v:
        p <<= 1
    return p


def robin_order(keys):
    """Independent pure-python robin-hood (linear probing, power-of-two, reserve(n)) emulation.

    keys: (n,3) int array in input order.  Returns the list of first-occurrence input indices in
    bucket-iteration order - what VoxelDownsampleSemantic emits (F/Preprocessing.cpp:73-91).
    """
    n = len(keys)
    if n == 0:
        return []
    B = pow2ceil(2 * n)
    mask = B - 1
    slot = [None] * B  # (dist, key tuple, idx)
    seen = set()
    for i in range(n):
        k = tuple(int(v) for v in keys[i])
        if k in seen:
            continue
        seen.add(k)
        cur = (0, k, i)
        b = ref_hash(k) & mask
        while True:
            r = slot[b]
            if r is None:
                slot[b] = cur
                break
            if cur[0] > r[0]:
                slot[b], cur = cur, r
            b = (b + 1) & mask
            cur = (cur[0] + 1, cur[1], cur[2])
    return [s[2] for s in slot if s is not None]


def pose_err(a, b):
    """translation / rotation-angle distance between two [qx,qy,qz,qw,tx,ty,tz] poses."""
    from scipy.spatial.transform import Rotation as R
    ra, rb = R.from_quat(a[:3 + 1]), R.from_quat(b[:4])
    dr = (ra.inv() * rb).magnitude()
    dt = np.linalg.norm(ra.inv().apply(np.asarray(b[4:]) - np.asarray(a[4:])))
    return dt, dr


def pose_matrix(p):
    from scipy.spatial.transform import Rotation as R
    T = np.eye(4)
    T[:3, :3] = R.from_quat(p[:4]).as_matrix()
    T[:3, 3] = p[4:]
    return T


def cluster_reference(xyz, dA, dR, dP):
    """Appendix C of SURVEY.md (F/SemanticCluster.cpp:22-174) in plain python for ONE class."""
    import math
    n = len(xyz)
    min_az = -24.8 * math.pi / 180
    max_az = 2 * math.pi / 180
    apr = []
    minr, maxr = 1.7976931348623157e308, 2.2250738585072014e-308
    for x, y, z in xyz:
        if x == 0 and y == 0:
            pol = 0.0
        elif y >= 0:
            pol = math.atan2(y, x)
        else:
            pol = math.atan2(y, x) + 2 * math.pi
        r = math.sqrt(x * x + y * y)
        az = math.atan2(z, r)
        minr = min(minr, r)
        maxr = max(maxr, r)
        apr.append((az, pol, r))
    length = int((maxr - minr) / dR) + 1
    width = int(round(360 / dP))
    height = int(((max_az - min_az) * 180 / math.pi) / dA) + 1
    cells = {}
    idx3 = []
    for i, (az, pol, r) in enumerate(apr):
        a = int(((az - min_az) * 180 / math.pi) / dA)
        p = int(pol * 180 / math.pi / dP)
        g = int((r - minr) / dR)
        idx3.append((p, g, a))
        cells.setdefault((p * length + g) + a * length * width, []).append(i)
    ci = [-1] * n
    cur = 0
    for i in range(n):
        if ci[i] != -1:
            continue
        p, g, a = idx3[i]
        neigh = []
        for z in range(a - 1, a + 2):
            if z < 0 or z > height - 1:
                continue
            for y in range(g - 1, g + 2):
                if y < 0 or y > length - 1:
                    continue
                for x in range(p - 1, p + 2):
                    px = x
                    if x < 0:
                        px = width - 1
                    if x > width - 1:
                        px = 0
                    neigh.extend(cells.get((px * length + y) + z * length * width, []))
        for j in neigh:
            oc, nc = ci[i], ci[j]
            if oc != -1 and nc != -1:
                if oc != nc:
                    ci = [nc if c == oc else c for c in ci]
            elif nc != -1:
                ci[i] = nc
            elif oc != -1:
                ci[j] = oc
        if ci[i] == -1:
            cur += 1
            ci[i] = cur
            for j in neigh:
                ci[j] = cur
    return ci
