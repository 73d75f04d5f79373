"""Which stage of the path overlaps with itself when several contexts share one GPU?  Runs one granular entry point
N times from T threads (one context each) and prints the aggregate call rate for T = 1, 2, 4."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from sg_slam_amd import api, synth  # noqa: E402


def xyzl(x, l):
    return np.concatenate([x, np.asarray(l, dtype=np.float64)[:, None]], axis=1)


def main():
    T_max = 4
    ctxs = [api.Context(api.preset("kitti", max_points=1 << 17, map_capacity_log2=19)) for _ in range(T_max)]
    scans = [synth.scan(k, "hdl64") for k in range(6)]
    for c in ctxs:
        for k in range(5):
            c.mainProcess(scans[k][0][:, :3].astype(np.float64), scans[k][2])
    xyz = scans[5][0][:, :3].astype(np.float64); lab = scans[5][2]
    px, pl = ctxs[0].PreprocessSemantic(xyz, lab, 100.0, 5.0)
    dx, dl = ctxs[0].VoxelDownsampleSemantic(px, pl, 0.5)[:2]
    sx, sl = ctxs[0].VoxelDownsampleSemantic(dx, dl, 1.5)[:2]
    cx, cl = ctxs[0].VoxelDownsampleSemantic(px, pl, 0.2)[:2]
    src = xyzl(sx, sl)
    guess = synth.gt_pose(5)
    jobs = {
        "register_frame (ICP loop, %d sources)" % len(src): lambda c: c.RegisterFrameSemantic(src, guess, 1.0, 0.1),
        "voxel_downsample 0.5 m (%d points)" % len(px): lambda c: c.VoxelDownsampleSemantic(px, pl, 0.5),
        "cluster_points (%d points)" % len(cx): lambda c: c.ClusterPoints(cx, cl, 2.0, 0.5, 2.0),
    }
    for name, fn in jobs.items():
        for T in (1, 2, 4):
            N = 60
            bar = threading.Barrier(T + 1)

            def run(k):
                fn(ctxs[k])
                bar.wait()
                for _ in range(N):
                    fn(ctxs[k])
                bar.wait()
            th = [threading.Thread(target=run, args=(k,)) for k in range(T)]
            [t.start() for t in th]
            bar.wait(); t0 = time.perf_counter(); bar.wait(); el = time.perf_counter() - t0
            [t.join() for t in th]
            print(f"{name:48s} threads {T}: {T * N / el:9.1f} calls/s  ({1e3 * el / N:.3f} ms per call per thread)", flush=True)


if __name__ == "__main__":
    main()
