"""Experiment: S independent scan streams (contexts + host threads) on ONE GPU."""
import os, sys, threading, time
if len(sys.argv) > 3:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sg_slam_amd import api, synth

S_list = [int(x) for x in sys.argv[1].split(",")]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W = 10
graph = os.environ.get("NOGRAPH") is None
dev = torch.device("cuda:0")
scans, labels = [], []
for k in range(W + K):
    xyzi, raw, lab = synth.scan(k, "hdl64")
    scans.append(torch.from_numpy(xyzi).to(dev)); labels.append(torch.from_numpy(lab).to(dev))
torch.cuda.synchronize()
n = scans[0].shape[0]
for S in S_list:
    ctxs = [api.Context(api.preset("kitti", max_points=1 << 17, map_capacity_log2=19, enable_graph=1 if graph else 0)) for _ in range(S)]
    bar = threading.Barrier(S + 1)
    def worker(c):
        res = api.SgsScanResult()
        for i in range(W):
            c.process_scan_dev(scans[i].data_ptr(), labels[i].data_ptr(), n, None, "", res)
        bar.wait()
        for i in range(W, W + K):
            c.process_scan_dev(scans[i].data_ptr(), labels[i].data_ptr(), n, None, "", res)
        bar.wait()
    th = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    print(f"streams {S}: {S*K/dt:.1f} scans/s  ({1e3*dt/K:.3f} ms per step of {S} scans)", flush=True)
    for c in ctxs: c.close()
