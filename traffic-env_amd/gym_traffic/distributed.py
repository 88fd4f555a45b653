"""Env-sharded multi-GPU rollouts: one process per GPU, no collective in the step.

Envs share nothing (each reference TrafficEnv owns its arrays, traffic_env.py:361-382), so rank k
simply owns the contiguous env-id range `shard_range(total, k, world)`; per-env inputs are functions
of the GLOBAL env id (spawner seeds, light-cycle offsets), which makes results independent of the
sharding.  The only exchange is optional: `RolloutGather` collects (obs, reward, done) snapshots on
rank 0 - ONE gather per snapshot over RCCL/xGMI (backend "nccl") on a side stream so the next ticks
overlap it, or over gloo on CPU tensors in tests.  Snapshots are double-buffered: starting snapshot
k waits only for snapshot k-2 (the previous user of the same buffer), never for k-1.
"""
import time

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) of `total` items for `rank`; sizes differ by at most one."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class RolloutGather(object):
    """Snapshot layout (int32 words, one contiguous buffer per rank and per slot):
        [ obs  n_max*obs_len | rewards (f32 bits) n_max*I | done  n_max ]
    with n_max the largest shard, so every rank sends the same number of words."""

    DEPTH = 2

    def __init__(self, n_local, obs_len, n_intersections, device, dst=0, group=None, counts=None,
                 single_rank_collective=False):
        """counts: envs per rank when the shards differ (`[hi - lo for lo, hi in (shard_range(total, r,
        world) ...)]` - a pure function of the sharding, so no exchange is needed); default: every
        rank holds n_local envs.
        single_rank_collective: issue the gather even in a process group of ONE rank (by default a single
        rank gathers nothing) - runs the backend's whole path (RCCL: communicator, side stream, work
        handles) on a one-GPU box."""
        self.dst, self.group = dst, group
        on = dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.device = torch.device(device)
        # gloo cannot gather device tensors: rehearsals on one GPU (and CPU tests) stage through host
        # memory; with RCCL ("nccl") the snapshots stay on the device
        backend = dist.get_backend(group) if on else "none"
        self.stage_dev = torch.device("cpu") if (backend == "gloo" and self.device.type == "cuda") else self.device
        self.n, self.L, self.I = int(n_local), int(obs_len), int(n_intersections)
        self.counts = [int(c) for c in counts] if counts is not None else [self.n] * self.world
        if len(self.counts) != self.world or self.counts[self.rank] != self.n:
            raise ValueError("counts must list every rank's envs, this rank's being n_local")
        self.n_max = max(self.counts)
        self.o_rew = self.n_max * self.L
        self.o_done = self.o_rew + self.n_max * self.I
        self.words = self.o_done + self.n_max
        pin = self.stage_dev.type == "cpu" and self.device.type == "cuda"
        self.snap = [self._buf(pin) for _ in range(self.DEPTH)]
        self.recv = None
        self.collect = self.world > 1 or (on and bool(single_rank_collective))
        if self.rank == dst and self.collect:
            self.recv = [[self._buf(False) for _ in range(self.world)] for _ in range(self.DEPTH)]
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.pending = [None] * self.DEPTH
        self.started = 0
        self.collectives = 0          # gathers issued (one per snapshot)
        self._timing = []             # per gather: (start event, end event) on the side stream, or seconds on the host
        self._host_s = 0.0

    def _buf(self, pin):
        t = torch.zeros((self.words,), dtype=torch.int32, device=self.stage_dev)
        return t.pin_memory() if pin else t

    def _wait_slot(self, k):
        w = self.pending[k]
        if w is not None:
            t0 = time.perf_counter()
            w.wait()
            if not self._timing:
                self._host_s += time.perf_counter() - t0      # (host-staged gathers: the time the caller spent in them)
            self.pending[k] = None

    def ms_per_snapshot(self):
        """Mean duration of one gather in milliseconds: on the side stream between the events around the collective
        (RCCL), or the host time spent issuing and waiting for it (gloo); None before the first gather."""
        if not self.collectives:
            return None
        self.wait()
        if self._timing:
            return sum(a.elapsed_time(b) for a, b in self._timing) / len(self._timing)
        return self._host_s / self.collectives * 1e3

    def start(self, obs, rewards, done):
        """Snapshot the three tensors and start gathering them to `dst`; returns immediately.  Only
        the gather that last used this snapshot's buffer (two starts ago) is waited for.

        The snapshot copy runs on the CALLER's stream, in order with the ticks before and after it: the
        kernels of the next ticks cannot overwrite obs / rewards / done under a copy that is still in
        flight (it is ~50 MB at cfg3, tens of microseconds).  Only the gather goes to the side stream, which
        waits for the copy; the caller's stream never waits for the gather of this snapshot."""
        k = self.started % self.DEPTH
        self.started += 1
        snap = self.snap[k]
        n, L, I = self.n, self.L, self.I
        self._wait_slot(k)            # (RCCL: orders the caller's stream after that gather; gloo: blocks)
        snap[:n * L].view(n, L).copy_(obs, non_blocking=True)
        snap[self.o_rew:self.o_rew + n * I].view(torch.float32).view(n, I).copy_(rewards, non_blocking=True)
        snap[self.o_done:self.o_done + n].copy_(done, non_blocking=True)       # u8 -> i32
        if self.side is None:
            ctx = _Null()
        elif self.stage_dev.type == "cpu":
            torch.cuda.current_stream(self.device).synchronize()    # host copies must have landed before gloo reads them
            ctx = _Null()
        else:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            ctx = torch.cuda.stream(self.side)
        if self.collect:
            timed_dev = self.side is not None and self.stage_dev.type != "cpu" and len(self._timing) < 256
            t0 = time.perf_counter()
            with ctx:
                if timed_dev:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.side)
                self.pending[k] = dist.gather(snap, self.recv[k] if self.rank == self.dst else None,
                                              dst=self.dst, group=self.group, async_op=True)
                if timed_dev:
                    e1.record(self.side)
                    self._timing.append((e0, e1))
            if not timed_dev:
                self._host_s += time.perf_counter() - t0
            self.collectives += 1

    def wait(self):
        """Block the host until every started gather has landed (no-op if none)."""
        for k in range(self.DEPTH):
            self._wait_slot(k)
        if self.side is not None:
            self.side.synchronize()

    def result(self):
        """On dst: (obs [sum n, L], rewards [sum n, I], done [sum n]) of the LAST started snapshot, in
        env-id order; None on the other ranks."""
        self.wait()
        k = (self.started - 1) % self.DEPTH
        if not self.collect:
            parts, counts = [self.snap[k]], [self.n]
        elif self.rank != self.dst:
            return None
        else:
            parts, counts = self.recv[k], self.counts
        L, I = self.L, self.I
        obs = torch.cat([p[:n * L].view(n, L) for p, n in zip(parts, counts)], dim=0)
        rew = torch.cat([p[self.o_rew:self.o_rew + n * I].view(torch.float32).view(n, I)
                         for p, n in zip(parts, counts)], dim=0)
        done = torch.cat([p[self.o_done:self.o_done + n] for p, n in zip(parts, counts)], dim=0).to(torch.uint8)
        return obs, rew, done


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
