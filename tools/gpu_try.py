"""Exploratory GPU-vs-oracle comparison (development aid; the real checks live in tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
from sg_slam_amd import api, synth

def pose_err(a, b):
    d = O.se3_mul(O.se3_inv(a), b)
    return np.linalg.norm(d[4:]), 2 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3]))

shape = sys.argv[1] if len(sys.argv) > 1 else "small"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = api.preset("kitti", max_points=1 << 18, map_capacity_log2=17)
ctx = api.Context(cfg)
xyzi, raw, lab = synth.scan(3, shape)
xyz = xyzi[:, :3].astype(np.float64)
print("n", len(xyz))
# keys
k_g = ctx.voxel_keys(xyz, 0.2); k_o = O.voxel_keys(xyz, 0.2)
print("keys equal", np.array_equal(k_g, k_o))
# preprocess
px_g, pl_g = ctx.PreprocessSemantic(xyz, lab, 100.0, 5.0); px_o, pl_o = O.preprocess(xyz, lab, 100.0, 5.0)
print("preprocess equal", px_g.shape, np.array_equal(px_g, px_o), np.array_equal(pl_g, pl_o))
kc_g = ctx.CorrectKITTIScan(xyz); kc_o = O.correct_kitti(xyz)
print("kitti max abs diff", np.abs(kc_g - kc_o).max(), "bit equal", np.array_equal(kc_g, kc_o))
ts = synth.mulran_timestamps(len(xyz))
ds_g = ctx.DeSkewScan(xyz, ts, synth.gt_pose(20), synth.gt_pose(21)); ds_o = O.deskew(xyz, ts, synth.gt_pose(20), synth.gt_pose(21))
print("deskew max abs diff", np.abs(ds_g - ds_o).max())
for v in (0.2, 0.5, 1.5):
    t = time.time(); gx, gl, gi = ctx.VoxelDownsampleSemantic(px_o, pl_o, v); tg = time.time() - t
    ox, ol, oi, info = O.voxel_downsample(px_o, pl_o, v)
    print("downsample", v, len(gi), len(oi), "index equal", np.array_equal(gi, oi), "xyz equal", np.array_equal(gx, ox), info, "%.1fms" % (tg * 1e3))
# map + correspondences + icp vs oracle, standalone
om = O.Map(1.0, 100.0, 20)
dx, dl, _, _ = O.voxel_downsample(px_o, pl_o, 0.5)
xyzl = np.concatenate([dx, dl[:, None].astype(np.float64)], axis=1)
pose0 = synth.gt_pose(3)
om.update(xyzl, pose0); ctx.Update(xyzl, pose0)
print("map size", om.size(), ctx.map_size())
ko, co, po = om.dump(); kg, cg, pg = ctx.map_dump()
print("map keys equal", np.array_equal(ko, kg), "counts equal", np.array_equal(co, cg), "pts max diff", np.abs(po - pg).max())
sx, sl, _, _ = O.voxel_downsample(dx, dl, 1.5)
src = np.concatenate([sx, sl[:, None].astype(np.float64)], axis=1)
guess = O.se3_mul(pose0, O.se3_exp([0.3, 0.05, 0.0, 0, 0, 0.01]))
srcw = np.concatenate([O.se3_act(guess, sx), sl[:, None].astype(np.float64)], axis=1)
so, to, qo, cand = om.get_correspondences(srcw, 6.0); sg, tg_, qg = ctx.GetCorrespondences(srcw, 6.0)
print("corr", len(qo), len(qg), "query idx equal", np.array_equal(qo, qg), "tgt max diff", np.abs(to - tg_).max() if len(qo) == len(qg) else None)
lo, mo = om.build_linear_system(srcw, 6.0, 0.667); lg, mg = ctx.BuildLinearSystem(srcw, 6.0, 0.667)
print("linsys n", mo, mg, "max rel diff", np.max(np.abs(lo - lg) / (np.abs(lo) + 1e-9)))
t = time.time(); pg_, itg = ctx.RegisterFrameSemantic(src, guess, 6.0, 0.667); tgpu = time.time() - t
t = time.time(); po_, info = om.register_frame(src, guess, 6.0, 0.667); tcpu = time.time() - t
print("icp iters", info["iterations"], itg, "pose err", pose_err(po_, pg_), "gpu %.1fms cpu %.1fms" % (tgpu * 1e3, tcpu * 1e3))
# cluster + graph
cx, cl, _, _ = O.voxel_downsample(px_o, pl_o, 0.2)
t = time.time(); bo, pco, nfo, nbo = O.cluster_points(cx, cl, 2.0, 0.5, 2.0); tc = time.time() - t
t = time.time(); bg, pcg, nfg, nbg = ctx.ClusterPoints(cx, cl, 2.0, 0.5, 2.0); tgc = time.time() - t
print("cluster boxes", len(bo), len(bg), "fg/bg", (nfo, nbo), (nfg, nbg), "pc equal", np.array_equal(pco, pcg),
      "boxes equal", bo.shape == bg.shape and np.array_equal(bo, bg), "cpu %.1fms gpu %.1fms" % (tc * 1e3, tgc * 1e3))
go = O.build_graph(bo); gg = ctx.BuildGraph(bo)
D = 90
print("graph hist equal", np.array_equal(go["desc"][:, :D], gg["desc"][:, :D]), "edges equal", np.array_equal(go["edges"], gg["edges"]),
      "edge_matrix max diff", np.abs(go["edge_matrix"] - gg["edge_matrix"]).max(), "evals max diff", np.abs(go["eigenvalues"] - gg["eigenvalues"]).max(),
      "tri equal", np.array_equal(go["triangles"], gg["triangles"]), "spectral max diff", np.abs(go["desc"][:, D:] - gg["desc"][:, D:]).max())
# full pipeline
ctx2 = api.Context(cfg); slam = O.Slam(O.kitti_config())
for k in range(nframes):
    xyzi, raw, lab = synth.scan(k, shape)
    t = time.time(); rg = ctx2.process_scan_f32(xyzi, lab); tg = time.time() - t
    t = time.time(); ro = slam.process(xyzi[:, :3].astype(np.float64), lab); to_ = time.time() - t
    et, er = pose_err(ro["pose"], rg["pose"])
    same = all(rg[a] == ro[a] for a in ("n_in", "n_cluster", "n_ds", "n_src"))
    print(k, "counts same", same, (rg["n_in"], rg["n_cluster"], rg["n_ds"], rg["n_src"]), "iters", ro["iterations"], rg["iterations"],
          "boxes", ro["n_boxes"], rg["n_boxes"], "map", ro["map_voxels"], rg["map_voxels"], "pose err %.2e m %.2e rad" % (et, er),
          "gpu %.2fms cpu %.2fms" % (tg * 1e3, to_ * 1e3))
s = ctx2.stats()
print("stats scans", s.scans, "cand", s.icp_candidates_total, "probes", s.icp_probes_total, "queries", s.icp_queries_total)
