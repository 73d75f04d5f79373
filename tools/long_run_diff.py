#!/usr/bin/env python3
"""Per-frame pose difference between the library and the oracle over a long synthetic drive (diagnostic).
usage: python tools/long_run_diff.py [frames] [shape]   -> gpurun_out/long_run_diff.csv"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from helpers import pose_err  # noqa: E402
from sg_slam_amd import api, synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
shape = sys.argv[2] if len(sys.argv) > 2 else "hdl64"
ctx = api.Context(api.preset("kitti", max_points=1 << 17, map_capacity_log2=17))
slam = O.Slam(O.kitti_config())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "long_run_diff.csv"), "w") as fh:
    fh.write("frame,dt_m,dr_rad,it_gpu,it_cpu,corr_gpu,corr_cpu,sigma_gpu,sigma_cpu,nsrc\n")
    for k in range(frames):
        xyzi, raw, lab = synth.scan(k, shape)
        rg = ctx.process_scan_f32(xyzi, lab, None, "")
        ro = slam.process(xyzi[:, :3].astype(np.float64), lab)
        et, er = pose_err(ro["pose"], rg["pose"])
        fh.write(f"{k},{et:.3e},{er:.3e},{rg['iterations']},{ro['iterations']},{rg['correspondences']},{ro['correspondences']},"
                 f"{rg['sigma']:.9f},{ro['sigma']:.9f},{rg['n_src']}\n")
print("done")
