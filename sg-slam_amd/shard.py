"""Host-side mirror of the key sharding of the voxel map (csrc/sgs_math.h: shard_owner / shard_stores;
include/sgs.h: sgs_shard_set).  The reference keeps one map in one address space (frontend/VoxelHashMap.hpp);
this partition is what lets BASELINE.json configs[4] (a dense map larger than one GPU wants to hold) spread it.

  owner(k)  = ((kx >> b) * 73856093 ^ (ky >> b) * 19349663) mod world      (uint32 arithmetic)
  stores(k) = any of the 9 voxels (kx+dx, ky+dy, kz), dx,dy in {-1,0,1}, is owned by the rank

A query is answered by the rank that owns its centre voxel; that rank stores all 27 neighbour voxels, so
the nearest neighbour it finds is the one the unsharded map finds, and no other rank answers the query.
"""
import numpy as np

_P1, _P2 = np.uint32(73856093), np.uint32(19349663)


def shard_owner(keys, block_log2, world):
    """keys: (n, 3) int32 voxel keys -> (n,) int32 owning rank."""
    k = np.asarray(keys, dtype=np.int32).reshape(-1, 3)
    bx = (k[:, 0] >> block_log2).astype(np.uint32)
    by = (k[:, 1] >> block_log2).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = (bx * _P1) ^ (by * _P2)
    return (h % np.uint32(world)).astype(np.int32)


def shard_stores(keys, block_log2, world, rank):
    """(n,) bool: the rank keeps this voxel (owns it or a voxel beside it in x/y)."""
    k = np.asarray(keys, dtype=np.int32).reshape(-1, 3)
    keep = np.zeros(len(k), dtype=bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            keep |= shard_owner(k + np.array([dx, dy, 0], dtype=np.int32), block_log2, world) == rank
    return keep


def torch_allreduce(a):
    """In-place SUM of a float64 numpy array over the default torch.distributed group (gloo or RCCL via a
    device bounce is the caller's choice; 28 doubles per Gauss-Newton iteration)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(a)
    if dist.get_backend() == "nccl":
        d = t.cuda()
        dist.all_reduce(d)
        t.copy_(d.cpu())
    else:
        dist.all_reduce(t)


class TorchAllReduce:
    """The same, with the device bounce buffer of the RCCL case allocated once (28 doubles per call)."""

    def __init__(self, local_rank=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.dev = torch.empty(32, dtype=torch.float64, device=f"cuda:{local_rank}") if dist.get_backend() == "nccl" else None

    def __call__(self, a):
        t = self.torch.from_numpy(a)
        if self.dev is None:
            self.dist.all_reduce(t)
        else:
            d = self.dev[:len(a)]
            d.copy_(t)
            self.dist.all_reduce(d)
            t.copy_(d)  # blocking D2H: the sums are in `a` when this returns
