"""Sequence-parallel multi-GPU runner helpers (mode "sequence-parallel").

A LiDAR sequence is strictly serial in time (the map at t depends on the pose at t), but sequences
share nothing, so the path shards as ONE PROCESS PER GPU, each running its own scan streams with no
data-path collective.  torch.distributed (RCCL on GPUs, gloo in CPU tests) is used only for the
barrier and the max-over-ranks timing the bench contract asks for.
"""
import os
import time


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def assign_sequences(lengths, world_size):
    """Deal sequences (given by their scan counts) to ranks, longest first onto the least loaded rank.

    Returns a list of index lists, one per rank.  KITTI 00-10 -> 11 sequences over N GPUs
    (BASELINE.json configs[3])."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out


class Dist:
    """Thin wrapper: a no-op for one process, torch.distributed otherwise."""

    def __init__(self, backend=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.pg = None
        self.backend = None
        if self.world > 1:
            import torch.distributed as dist
            if backend is None:
                import torch
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.backend = backend
            if not dist.is_initialized():
                dist.init_process_group(backend=backend)
            self.pg = dist

    def _tensor(self, v):
        import torch
        dev = f"cuda:{self.local_rank}" if self.backend == "nccl" else "cpu"
        return torch.tensor([float(v)], dtype=torch.float64, device=dev)

    def barrier(self):
        if self.pg is not None:
            if self.backend == "nccl":
                import torch
                self.pg.barrier(device_ids=[self.local_rank])
                torch.cuda.synchronize()
            else:
                self.pg.barrier()

    def max(self, v):
        if self.pg is None:
            return float(v)
        t = self._tensor(v)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return float(t.item())

    def sum(self, v):
        if self.pg is None:
            return float(v)
        t = self._tensor(v)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.pg is not None and self.pg.is_initialized():
            self.pg.destroy_process_group()


def timed_steps(dist, step_fn, steps, sync_fn=None):
    """barrier + device sync, EXACTLY `steps` calls of step_fn(i), device sync + barrier; returns the
    max-over-ranks wall time in seconds and the whole-job number of steps."""
    if sync_fn:
        sync_fn()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    if sync_fn:
        sync_fn()
    elapsed = time.perf_counter() - t0
    dist.barrier()
    return dist.max(elapsed), int(round(dist.sum(steps)))
