"""One-rank NCCL (RCCL) rehearsal of the barrier / max / sum calls bench.py makes at --gpus N (run on a GPU box)."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29511')
import torch, torch.distributed as dist
from sg_slam_amd.runner import Dist, timed_steps
torch.cuda.set_device(0)
d = Dist()
assert d.pg is None
dist.init_process_group(backend='nccl')
d.backend, d.pg = 'nccl', dist
d.barrier()
print('max', d.max(1.5), 'sum', d.sum(2))
print(timed_steps(d, lambda i: None, 3, sync_fn=torch.cuda.synchronize))
d.close()
print('nccl path ok')
