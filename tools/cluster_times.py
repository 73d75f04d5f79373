"""Development aid: wall time of sgs_cluster_points (upload + routing + clustering + boxes, synchronous) per scan,
on the 0.2 m cloud the library's own crop + downsample produce."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sg_slam_amd import api, synth  # noqa: E402

ctx = api.Context(api.preset("kitti", max_points=1 << 17), device=0)
for k in range(24):
    xyzi, raw, _ = synth.scan(k, sys.argv[1] if len(sys.argv) > 1 else "hdl64")
    lab = api.remap_labels(raw)
    px, pl = ctx.PreprocessSemantic(xyzi[:, :3].astype(np.float64), lab, 100.0, 5.0)
    cx, cl = ctx.VoxelDownsampleSemantic(px, pl, 0.2)[:2]
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        b, pc, nfg, nbg = ctx.ClusterPoints(cx, cl, 2.0, 0.5, 2.0)
        ts.append((time.perf_counter() - t0) * 1e6)
    print(k, len(cx), nfg, len(b), "us:", " ".join(f"{t:6.0f}" for t in ts), flush=True)
ctx.close()
