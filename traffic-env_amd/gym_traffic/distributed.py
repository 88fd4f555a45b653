"""Env-sharded multi-GPU rollouts: one process per GPU, no collective in the step.

Envs share nothing (each reference TrafficEnv owns its arrays, traffic_env.py:361-382), so rank k
simply owns the contiguous env-id range `shard_range(total, k, world)`; per-env inputs are functions
of the GLOBAL env id (spawner seeds, light-cycle offsets), which makes results independent of the
sharding.  The only exchange is optional: `RolloutGather` collects (obs, reward, done) snapshots on
rank 0 - a gather over RCCL/xGMI (backend "nccl") on a side stream so the next ticks overlap it, or
over gloo on CPU tensors in tests.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) of `total` items for `rank`; sizes differ by at most one."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class RolloutGather(object):
    def __init__(self, n_local, obs_len, n_intersections, device, dst=0, group=None):
        self.dst, self.group = dst, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device(device)
        # gloo cannot gather device tensors: rehearsals on one GPU (and CPU tests) stage through host
        # memory; with RCCL ("nccl") the snapshots stay on the device
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.stage_dev = torch.device("cpu") if (backend == "gloo" and self.device.type == "cuda") else self.device
        self.shapes = ((n_local, obs_len), (n_local, n_intersections), (n_local,))
        self.dtypes = (torch.int32, torch.float32, torch.uint8)
        # snapshots: obs/rewards are live buffers the next tick overwrites
        self.snap = [torch.empty(s, dtype=t, device=self.stage_dev) for s, t in zip(self.shapes, self.dtypes)]
        self.recv = None
        if self.rank == dst and self.world > 1:
            self.recv = [[torch.empty(s, dtype=t, device=self.stage_dev) for _ in range(self.world)]
                         for s, t in zip(self.shapes, self.dtypes)]
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.pending = []

    def start(self, obs, rewards, done):
        """Snapshot the three tensors and start gathering them to `dst`; returns immediately."""
        self.wait()
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            ctx = torch.cuda.stream(self.side)
        else:
            ctx = _Null()
        with ctx:
            for s, t in zip(self.snap, (obs, rewards, done)):
                s.copy_(t, non_blocking=True)
            if self.stage_dev.type == "cpu" and self.side is not None:
                self.side.synchronize()            # host copies must have landed before gloo reads them
            if self.world > 1:
                for k, s in enumerate(self.snap):
                    self.pending.append(dist.gather(s, self.recv[k] if self.rank == self.dst else None,
                                                    dst=self.dst, group=self.group, async_op=True))

    def wait(self):
        """Block the host until the last started gather has landed (no-op if none)."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.side is not None:
            self.side.synchronize()

    def result(self):
        """On dst: (obs [world*n, L], rewards [world*n, I], done [world*n]) in env-id order."""
        self.wait()
        if self.world == 1:
            return tuple(self.snap)
        if self.rank != self.dst:
            return None
        return tuple(torch.cat(parts, dim=0) for parts in self.recv)


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
