"""ORACLE (test infrastructure, NOT product code): ctypes wrapper over oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: the reference ships no tests or golden vectors and cannot be built here
(see oracle/sgo_core.hpp).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_vp = ctypes.c_void_p
c_i64 = ctypes.c_int64


class SgoConfig(ctypes.Structure):
    _fields_ = [
        ("voxel_size", c_dbl), ("voxel_size_cluster", c_dbl), ("max_range", c_dbl), ("min_range", c_dbl),
        ("max_points_per_voxel", ctypes.c_int32),
        ("min_motion_th", c_dbl), ("initial_threshold", c_dbl),
        ("model_deviation_trans", c_dbl), ("model_deviation_rot", c_dbl), ("inlier_rate_th", c_dbl),
        ("deskew", ctypes.c_int32), ("relocalization_enable", ctypes.c_int32),
        ("deltaA", c_dbl), ("deltaR", c_dbl), ("deltaP", c_dbl),
        ("edge_dis_th", c_dbl), ("subgraph_edge_th", c_dbl),
        ("subinterval", ctypes.c_int32), ("graph_node_dimension", ctypes.c_int32),
        ("nn_vehicle", c_dbl), ("nn_pole", c_dbl), ("max_local_graph_map_range", c_dbl),
        ("order_mode", ctypes.c_int32), ("prune_mode", ctypes.c_int32),
        ("enable_graph", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


def kitti_config(**kw):
    """ros/launch/semgraph_slam_kitti.launch:12-60 effective values."""
    c = SgoConfig(voxel_size=1.0, voxel_size_cluster=0.2, max_range=100.0, min_range=5.0,
                  max_points_per_voxel=20, min_motion_th=0.1, initial_threshold=2.0,
                  model_deviation_trans=0.12, model_deviation_rot=0.01, inlier_rate_th=0.43,
                  deskew=0, relocalization_enable=0, deltaA=2.0, deltaR=0.5, deltaP=2.0,
                  edge_dis_th=60.0, subgraph_edge_th=20.0, subinterval=30, graph_node_dimension=30,
                  nn_vehicle=2.0, nn_pole=2.0, max_local_graph_map_range=100.0,
                  order_mode=0, prune_mode=0, enable_graph=1, reserved=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        L.sgo_map_create.restype = c_vp
        L.sgo_map_create.argtypes = [c_dbl, c_dbl, c_int, c_int]
        L.sgo_slam_create.restype = c_vp
        L.sgo_slam_create.argtypes = [ctypes.POINTER(SgoConfig)]
        L.sgo_slam_map.restype = c_vp
        L.sgo_slam_map.argtypes = [c_vp]
        L.sgo_voxel_hash.restype = ctypes.c_uint32
        for name in ("sgo_map_destroy", "sgo_map_clear", "sgo_slam_destroy"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [c_vp]
        L.sgo_map_empty.argtypes = [c_vp]
        L.sgo_map_add_points.restype = None
        L.sgo_map_add_points.argtypes = [c_vp, c_vp, c_int]
        L.sgo_map_update.restype = None
        L.sgo_map_update.argtypes = [c_vp, c_vp, c_int, c_vp]
        L.sgo_map_prune.restype = None
        L.sgo_map_prune.argtypes = [c_vp, c_vp]
        L.sgo_map_size.restype = None
        L.sgo_map_size.argtypes = [c_vp, c_vp, c_vp]
        L.sgo_map_pointcloud.argtypes = [c_vp, c_vp, c_int]
        L.sgo_map_dump.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp]
        L.sgo_map_get_correspondences.argtypes = [c_vp, c_vp, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp]
        L.sgo_build_linear_system.argtypes = [c_vp, c_vp, c_int, c_dbl, c_dbl, c_vp]
        L.sgo_register_frame.restype = None
        L.sgo_register_frame.argtypes = [c_vp, c_vp, c_int, c_vp, c_dbl, c_dbl, c_vp, c_vp]
        L.sgo_fast_icp.restype = None
        L.sgo_fast_icp.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, ctypes.c_float, c_int, c_dbl, c_vp, c_vp]
        L.sgo_scan_descriptor.argtypes = [c_vp, c_int, c_vp, c_vp, c_int, c_dbl, c_int, c_vp]
        L.sgo_fast_atan2.restype = c_dbl
        L.sgo_fast_atan2.argtypes = [c_dbl, c_dbl]
        L.sgo_graphmap_create.restype = c_vp
        L.sgo_graphmap_create.argtypes = [c_dbl, c_dbl, c_int, c_int, c_dbl, c_dbl, c_dbl]
        L.sgo_graphmap_destroy.restype = None
        L.sgo_graphmap_destroy.argtypes = [c_vp]
        L.sgo_graphmap_track.argtypes = [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]
        L.sgo_graphmap_instances.argtypes = [c_vp, c_vp, c_int]
        L.sgo_graphmap_relocalize.argtypes = [c_vp, c_vp, c_int, c_dbl, ctypes.c_uint32, c_vp, c_vp, c_vp]
        L.sgo_find_ins_match.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp]
        L.sgo_solve_svd.restype = None
        L.sgo_solve_svd.argtypes = [c_vp, c_vp, c_int, c_vp, c_vp]
        L.sgo_ransac_alignment.argtypes = [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, ctypes.c_uint32, c_vp, c_vp, c_vp]
        L.sgo_slam_graphmap_info.restype = None
        L.sgo_slam_graphmap_info.argtypes = [c_vp, c_vp]
        L.sgo_voxel_keys.restype = None
        L.sgo_voxel_keys.argtypes = [c_vp, c_int, c_dbl, c_vp]
        L.sgo_preprocess.argtypes = [c_vp, c_vp, c_int, c_dbl, c_dbl, c_vp, c_vp]
        L.sgo_correct_kitti.restype = None
        L.sgo_correct_kitti.argtypes = [c_vp, c_int, c_vp]
        L.sgo_deskew.restype = None
        L.sgo_deskew.argtypes = [c_vp, c_vp, c_int, c_vp, c_vp, c_vp]
        L.sgo_voxel_downsample.argtypes = [c_vp, c_vp, c_int, c_dbl, c_int, c_i64, c_vp, c_vp, c_vp, c_vp]
        L.sgo_cluster_points.argtypes = [c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_int, c_vp, c_vp, c_vp, c_vp]
        L.sgo_build_graph.argtypes = [c_vp, c_int, c_dbl, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_int,
                                      c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]
        L.sgo_slam_process.restype = None
        L.sgo_slam_process.argtypes = [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.sgo_slam_get_cloud.argtypes = [c_vp, c_int, c_vp, c_vp, c_int]
        L.sgo_slam_get_boxes.argtypes = [c_vp, c_vp, c_int]
        for name in ("sgo_se3_exp", "sgo_se3_log", "sgo_se3_inv"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [c_vp, c_vp]
        L.sgo_se3_mul.restype = None
        L.sgo_se3_mul.argtypes = [c_vp, c_vp, c_vp]
        L.sgo_se3_act.restype = None
        L.sgo_se3_act.argtypes = [c_vp, c_vp, c_int, c_vp]
        L.sgo_ldlt6_solve.restype = None
        L.sgo_ldlt6_solve.argtypes = [c_vp, c_vp, c_vp]
        L.sgo_sym_eig.restype = None
        L.sgo_sym_eig.argtypes = [c_int, c_vp, c_vp, c_vp]
        L.sgo_set_threads.restype = None
        L.sgo_set_threads.argtypes = [c_int]
        _LIB = L
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def set_threads(n):
    lib().sgo_set_threads(int(n))


def max_threads():
    return int(lib().sgo_get_max_threads())


# ---- SE3 ----
def se3_exp(t6):
    t6 = _f64(t6); out = np.empty(7)
    lib().sgo_se3_exp(t6.ctypes.data, out.ctypes.data)
    return out


def se3_log(p7):
    p7 = _f64(p7); out = np.empty(6)
    lib().sgo_se3_log(p7.ctypes.data, out.ctypes.data)
    return out


def se3_mul(a, b):
    a = _f64(a); b = _f64(b); out = np.empty(7)
    lib().sgo_se3_mul(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def se3_inv(a):
    a = _f64(a); out = np.empty(7)
    lib().sgo_se3_inv(a.ctypes.data, out.ctypes.data)
    return out


def se3_act(a, xyz):
    a = _f64(a); xyz = _f64(xyz); out = np.empty_like(xyz)
    lib().sgo_se3_act(a.ctypes.data, xyz.ctypes.data, len(xyz), out.ctypes.data)
    return out


def ldlt6_solve(A, b):
    A = _f64(A); b = _f64(b); x = np.empty(6)
    lib().sgo_ldlt6_solve(A.ctypes.data, b.ctypes.data, x.ctypes.data)
    return x


def sym_eig(A):
    A = _f64(A); n = A.shape[0]
    w = np.empty(n); v = np.empty((n, n))
    lib().sgo_sym_eig(n, A.ctypes.data, w.ctypes.data, v.ctypes.data)
    return w, v


# ---- stages ----
def voxel_keys(xyz, voxel):
    xyz = _f64(xyz); keys = np.empty((len(xyz), 3), dtype=np.int32)
    lib().sgo_voxel_keys(xyz.ctypes.data, len(xyz), float(voxel), keys.ctypes.data)
    return keys


def voxel_hash(k):
    return int(lib().sgo_voxel_hash(int(k[0]), int(k[1]), int(k[2])))


def preprocess(xyz, label, max_range, min_range):
    xyz = _f64(xyz); label = _i32(label); n = len(xyz)
    ox = np.empty((n, 3)); ol = np.empty(n, dtype=np.int32)
    m = lib().sgo_preprocess(xyz.ctypes.data, label.ctypes.data, n, float(max_range), float(min_range),
                             ox.ctypes.data, ol.ctypes.data)
    return ox[:m].copy(), ol[:m].copy()


def correct_kitti(xyz):
    xyz = _f64(xyz); out = np.empty_like(xyz)
    lib().sgo_correct_kitti(xyz.ctypes.data, len(xyz), out.ctypes.data)
    return out


def deskew(xyz, ts, start7, finish7):
    xyz = _f64(xyz); ts = _f64(ts); s = _f64(start7); f = _f64(finish7); out = np.empty_like(xyz)
    lib().sgo_deskew(xyz.ctypes.data, ts.ctypes.data, len(xyz), s.ctypes.data, f.ctypes.data, out.ctypes.data)
    return out


def voxel_downsample(xyz, label, voxel, order_mode=0, probe_limit=4096):
    """Returns (xyz, label, index, info) with info = dict(buckets, max_dist, grow_by_probe, wrapped)."""
    xyz = _f64(xyz); label = _i32(label); n = len(xyz)
    ox = np.empty((n, 3)); ol = np.empty(n, dtype=np.int32); oi = np.empty(n, dtype=np.int32)
    info = np.zeros(4, dtype=np.int64)
    m = lib().sgo_voxel_downsample(xyz.ctypes.data, label.ctypes.data, n, float(voxel), int(order_mode),
                                   int(probe_limit), ox.ctypes.data, ol.ctypes.data, oi.ctypes.data,
                                   info.ctypes.data)
    return ox[:m].copy(), ol[:m].copy(), oi[:m].copy(), dict(
        buckets=int(info[0]), max_dist=int(info[1]), grow_by_probe=int(info[2]), wrapped=int(info[3]))


class Map:
    def __init__(self, voxel=1.0, max_dist=100.0, cap=20, prune_mode=0, handle=None):
        self.cap = cap
        self._own = handle is None
        self.h = handle if handle is not None else lib().sgo_map_create(float(voxel), float(max_dist), int(cap), int(prune_mode))

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().sgo_map_destroy(self.h)
            self.h = None

    def clear(self):
        lib().sgo_map_clear(self.h)

    def empty(self):
        return bool(lib().sgo_map_empty(self.h))

    def add_points(self, xyzl):
        xyzl = _f64(xyzl)
        lib().sgo_map_add_points(self.h, xyzl.ctypes.data, len(xyzl))

    def update(self, xyzl, pose7):
        xyzl = _f64(xyzl); p = _f64(pose7)
        lib().sgo_map_update(self.h, xyzl.ctypes.data, len(xyzl), p.ctypes.data)

    def prune(self, origin):
        o = _f64(origin)
        lib().sgo_map_prune(self.h, o.ctypes.data)

    def size(self):
        v = c_int(); p = c_int()
        lib().sgo_map_size(self.h, ctypes.byref(v), ctypes.byref(p))
        return v.value, p.value

    def pointcloud(self):
        _, npts = self.size()
        out = np.empty((max(npts, 1), 4))
        n = lib().sgo_map_pointcloud(self.h, out.ctypes.data, npts)
        return out[:n].copy()

    def dump(self):
        nv, _ = self.size()
        keys = np.empty((max(nv, 1), 3), dtype=np.int32); cnt = np.empty(max(nv, 1), dtype=np.int32)
        pts = np.empty((max(nv, 1), self.cap, 4))
        n = lib().sgo_map_dump(self.h, nv, keys.ctypes.data, cnt.ctypes.data, pts.ctypes.data)
        return keys[:n].copy(), cnt[:n].copy(), pts[:n].copy()

    def get_correspondences(self, xyzl, max_dist):
        xyzl = _f64(xyzl); n = len(xyzl)
        src = np.empty((max(n, 1), 4)); tgt = np.empty((max(n, 1), 4)); qi = np.empty(max(n, 1), dtype=np.int32)
        cand = c_i64()
        m = lib().sgo_map_get_correspondences(self.h, xyzl.ctypes.data, n, float(max_dist), src.ctypes.data,
                                              tgt.ctypes.data, qi.ctypes.data, ctypes.byref(cand))
        return src[:m].copy(), tgt[:m].copy(), qi[:m].copy(), cand.value

    def build_linear_system(self, xyzl_world, max_dist, kernel):
        xyzl = _f64(xyzl_world); out = np.empty(27)
        m = lib().sgo_build_linear_system(self.h, xyzl.ctypes.data, len(xyzl), float(max_dist), float(kernel),
                                          out.ctypes.data)
        return out, m

    def register_frame(self, xyzl, guess7, max_corr, kernel):
        xyzl = _f64(xyzl); g = _f64(guess7); out = np.empty(7); info = np.zeros(3, dtype=np.int64)
        lib().sgo_register_frame(self.h, xyzl.ctypes.data, len(xyzl), g.ctypes.data, float(max_corr),
                                 float(kernel), out.ctypes.data, info.ctypes.data)
        return out, dict(iterations=int(info[0]), candidates=int(info[1]), correspondences=int(info[2]))


def fast_icp(xyzl_a, xyzl_b, init7, voxel_size_map=1.0, max_points_per_voxel=5, max_corr=0.5):
    """loopclosure/FastIcp.cpp:205-220 get_trans: map <- cloud B, register cloud A; -> (pose7, info)."""
    a = _f64(xyzl_a); b = _f64(xyzl_b); g = _f64(init7); out = np.empty(7); info = np.zeros(3, dtype=np.int64)
    lib().sgo_fast_icp(a.ctypes.data, len(a), b.ctypes.data, len(b), g.ctypes.data, float(voxel_size_map),
                       int(max_points_per_voxel), float(max_corr), out.ctypes.data, info.ctypes.data)
    return out, dict(iterations=int(info[0]), candidates=int(info[1]), correspondences=int(info[2]))


def find_ins_match(frame, gmap):
    """GraphMap.cpp:88-113 on two graphs given as dicts with boxes, desc, tri_counts, triangles."""
    def arrs(g):
        return (_f64(g["boxes"]).reshape(-1, 8), np.ascontiguousarray(g["desc"], dtype=np.float32), _i32(g["tri_counts"]),
                _f64(g["triangles"]).reshape(-1, 3))
    fb, fd, fc, ft = arrs(frame); mb, md, mc, mt = arrs(gmap)
    out = np.zeros((max(len(fb), 1), 3), dtype=np.int32)
    n = lib().sgo_find_ins_match(fb.ctypes.data, len(fb), fd.ctypes.data, fc.ctypes.data, ft.ctypes.data, mb.ctypes.data, len(mb),
                                 md.ctypes.data, mc.ctypes.data, mt.ctypes.data, fd.shape[1] if fd.ndim == 2 else 1, out.ctypes.data)
    return out[:n].copy()


def solve_svd(p1, p2):
    """core/CommonUtils.cpp:264-316 -> (R 3x3, t) with R @ p1 + t ~ p2."""
    a = _f64(p1); b = _f64(p2); R = np.empty((3, 3)); t = np.empty(3)
    lib().sgo_solve_svd(a.ctypes.data, b.ctypes.data, len(a), R.ctypes.data, t.ctypes.data)
    return R, t


def ransac_alignment(m1, m2, nodes1, nodes2, max_iter=500, seed=1):
    """core/CommonUtils.cpp:115-204 with a seeded engine -> (R, t, score, best_inlier)."""
    a = _f64(m1); b = _f64(m2); n1 = _f64(nodes1); n2 = _f64(nodes2); R = np.empty((3, 3)); t = np.empty(3); sc = c_dbl()
    best = lib().sgo_ransac_alignment(a.ctypes.data, b.ctypes.data, len(a), n1.ctypes.data, len(n1), n2.ctypes.data, len(n2),
                                      int(max_iter), int(seed), R.ctypes.data, t.ctypes.data, ctypes.byref(sc))
    return R, t, sc.value, best


class GraphMap:
    """frontend/GraphMap.{hpp,cpp}: the local map of tracked instances."""

    def __init__(self, edge_dis_th=60.0, subgraph_edge_th=20.0, subinterval=30, dim=30, nn_vehicle=2.0, nn_pole=2.0, max_distance=100.0):
        self.h = lib().sgo_graphmap_create(edge_dis_th, subgraph_edge_th, subinterval, dim, nn_vehicle, nn_pole, max_distance)

    def __del__(self):
        if getattr(self, "h", None):
            lib().sgo_graphmap_destroy(self.h)
            self.h = None

    def track(self, boxes, pose7):
        """BuildGraph -> FindInsMatch -> Update; -> (matches (nb, 3) [flag, frame id, map id], new_instance ids)."""
        b = _f64(boxes).reshape(-1, 8); p = _f64(pose7); nb = len(b)
        m = np.zeros((max(nb, 1), 3), dtype=np.int32); fresh = np.zeros(max(nb, 1), dtype=np.int32); nn = ctypes.c_int32()
        n = lib().sgo_graphmap_track(self.h, b.ctypes.data, nb, p.ctypes.data, m.ctypes.data, fresh.ctypes.data, ctypes.byref(nn))
        return m[:n].copy(), fresh[:nn.value].copy()

    def instances(self):
        n = lib().sgo_graphmap_instances(self.h, None, 0)
        out = np.empty((max(n, 1), 9))
        lib().sgo_graphmap_instances(self.h, out.ctypes.data, n)
        return out[:n].copy()

    def relocalize(self, boxes, inlier_rate_th=0.4, seed=1):
        b = _f64(boxes).reshape(-1, 8); pose = np.empty(7); sc = c_dbl(); nm = ctypes.c_int32()
        ok = lib().sgo_graphmap_relocalize(self.h, b.ctypes.data, len(b), float(inlier_rate_th), int(seed), pose.ctypes.data,
                                           ctypes.byref(sc), ctypes.byref(nm))
        return bool(ok), pose, sc.value, nm.value


def scan_descriptor(boxes, xyz, label, edge_dis_th=60.0, subinterval=30):
    """loopclosure/LoopClosure.cpp:15-32 GenScanDescriptors on the boxes + the 0.2 m cluster cloud of a scan."""
    boxes = _f64(boxes).reshape(-1, 8); xyz = _f64(xyz); label = _i32(label)
    out = np.empty(6 * subinterval + 3 + 48, dtype=np.float32)
    n = lib().sgo_scan_descriptor(boxes.ctypes.data, len(boxes), xyz.ctypes.data, label.ctypes.data, len(xyz),
                                  float(edge_dis_th), int(subinterval), out.ctypes.data)
    return out[:n]


def fast_atan2(y, x):
    return float(lib().sgo_fast_atan2(float(y), float(x)))


def cluster_points(xyz, label, dA, dR, dP, max_boxes=4096):
    xyz = _f64(xyz); label = _i32(label); n = len(xyz)
    boxes = np.empty((max_boxes, 8)); pc = np.empty(max(n, 1), dtype=np.int32)
    nfg = c_int(); nbg = c_int()
    nb = lib().sgo_cluster_points(xyz.ctypes.data, label.ctypes.data, n, float(dA), float(dR), float(dP),
                                  max_boxes, boxes.ctypes.data, pc.ctypes.data, ctypes.byref(nfg), ctypes.byref(nbg))
    return boxes[:nb].copy(), pc[:n].copy(), nfg.value, nbg.value


def build_graph(boxes, edge_dis_th=60.0, subinterval=30, dim=30, subgraph_edge_th=20.0):
    boxes = _f64(boxes).reshape(-1, 8); nb = len(boxes)
    D = 3 * subinterval + dim
    desc = np.zeros((max(nb, 1), D), dtype=np.float32); em = np.zeros((max(nb, 1), max(nb, 1)))
    ev = np.zeros(max(nb, 1)); max_e = max(nb * nb, 1)
    edges = np.zeros((max_e, 2), dtype=np.int32); evalue = np.zeros(max_e); ew = np.zeros(max_e)
    tc = np.zeros(max(nb, 1), dtype=np.int32); max_tri = max(nb * nb * nb, 1)
    tri = np.zeros((max_tri, 3)); c3 = np.zeros(3, dtype=np.int32)
    ne = lib().sgo_build_graph(boxes.ctypes.data, nb, float(edge_dis_th), int(subinterval), int(dim),
                               float(subgraph_edge_th), desc.ctypes.data, em.ctypes.data, ev.ctypes.data,
                               max_e, edges.ctypes.data, evalue.ctypes.data, ew.ctypes.data, tc.ctypes.data,
                               max_tri, tri.ctypes.data, c3.ctypes.data)
    nt = int(tc[:nb].sum())
    return dict(desc=desc[:nb], edge_matrix=em[:nb, :nb], eigenvalues=ev[:nb], edges=edges[:ne],
                edge_value=evalue[:ne], edge_weights=ew[:ne], tri_counts=tc[:nb], triangles=tri[:nt],
                counts=c3)


class Slam:
    def __init__(self, cfg=None):
        self.cfg = cfg if cfg is not None else kitti_config()
        self.h = lib().sgo_slam_create(ctypes.byref(self.cfg))

    def __del__(self):
        if getattr(self, "h", None):
            lib().sgo_slam_destroy(self.h)
            self.h = None

    def process(self, xyz, label, timestamps=None, is_kitti=False):
        xyz = _f64(xyz); label = _i32(label)
        ts = _f64(timestamps) if timestamps is not None else None
        pose = np.empty(7); sigma = c_dbl(); counts = np.zeros(8, dtype=np.int64); ms = np.zeros(8)
        cand = c_i64()
        lib().sgo_slam_process(self.h, xyz.ctypes.data, label.ctypes.data, len(xyz),
                               ts.ctypes.data if ts is not None else None, int(bool(is_kitti)),
                               pose.ctypes.data, ctypes.byref(sigma), counts.ctypes.data, ms.ctypes.data,
                               ctypes.byref(cand))
        return dict(pose=pose, sigma=sigma.value, n_in=int(counts[0]), n_cluster=int(counts[1]),
                    n_ds=int(counts[2]), n_src=int(counts[3]), n_boxes=int(counts[4]),
                    iterations=int(counts[5]), map_voxels=int(counts[6]), correspondences=int(counts[7]),
                    candidates=cand.value, stage_ms=ms)

    def graphmap_info(self):
        info = np.zeros(4, dtype=np.int64)
        lib().sgo_slam_graphmap_info(self.h, info.ctypes.data)
        return dict(instances=int(info[0]), attempts=int(info[1]), relocalizations=int(info[2]), relocalized=bool(info[3]))

    def cloud(self, which):
        n = lib().sgo_slam_get_cloud(self.h, which, None, None, 0)
        xyz = np.empty((max(n, 1), 3)); lab = np.empty(max(n, 1), dtype=np.int32)
        lib().sgo_slam_get_cloud(self.h, which, xyz.ctypes.data, lab.ctypes.data, n)
        return xyz[:n].copy(), lab[:n].copy()

    def boxes(self):
        n = lib().sgo_slam_get_boxes(self.h, None, 0)
        b = np.empty((max(n, 1), 8))
        lib().sgo_slam_get_boxes(self.h, b.ctypes.data, n)
        return b[:n].copy()

    def map(self):
        return Map(cap=self.cfg.max_points_per_voxel, handle=lib().sgo_slam_map(self.h))
