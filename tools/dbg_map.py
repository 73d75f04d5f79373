import sys; sys.path.insert(0,'.')
import numpy as np
from sg_slam_amd import api
ctx = api.Context(api.preset("kitti", max_points=1<<12, map_capacity_log2=10))
pts = np.concatenate([np.random.default_rng(0).random((100,3))*5, np.zeros((100,1))],axis=1)
ctx.AddPoints(pts); print("after add", ctx.map_size(), len(ctx.Pointcloud()))
ctx.RemovePointsFarFromLocation([0,0,0]); print("after prune", ctx.map_size())
