"""Multi-process (world_size 2, gloo, CPU) coverage of the env-sharded rollout path: shard
arithmetic, sharding-independent per-env inputs, the (obs, reward, done) gather to rank 0 and the
benchmark's cross-rank reduction.  The step itself needs a GPU and is covered by -m gpu tests; here
each rank's "step output" is produced by the oracle, which is a checker role."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gym_traffic.distributed import shard_range, RolloutGather
        from gym_traffic import workload as wl
        from gym_traffic.envs.roadgraph import GridRoad
        from oracle.oracle import OracleEnv

        total, m, n, L, C = 6, 2, 2, 120.0, 12
        lo, hi = shard_range(total, rank, world)
        E = hi - lo
        g = GridRoad(m, n, L)
        g.generate_entrypoints(0)
        env = OracleEnv(m, n, L, C, g.dest, g.phases, g.nexts, n_envs=E)
        env.reset(np.zeros(env.I, np.int32))
        sizes = [b - a for a, b in (shard_range(total, r, world) for r in range(world))]
        gather = RolloutGather(E, env.obs.shape[1], env.I, "cpu", counts=sizes)
        snaps = []
        for t in range(30):
            ids = np.arange(lo, hi)                               # GLOBAL env ids drive the inputs
            act = wl.cycle_actions(ids, env.I, t, period=5)
            roads = [wl.spawn_roads_for_tick(g.entrypoints, t + int(k), period=3) for k in ids]
            obs, rew, done = env.step(act, roads)
            if (t + 1) % 10 == 0:
                gather.start(torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done))
                res = gather.result()
                if rank == 0:
                    snaps.append([r.clone().numpy() for r in res])
                else:
                    assert res is None
        # the benchmark's reduction: max time over ranks, sum of updates
        tt = torch.tensor([1.0 + rank], dtype=torch.float64)
        uu = torch.tensor([float(env.vehicle_updates)], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(uu, op=dist.ReduceOp.SUM)
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), t_max=tt.numpy(), updates=uu.numpy(),
                     **{"s%d_%d" % (i, k): a for i, s in enumerate(snaps) for k, a in enumerate(s)})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _single_process_reference():
    from gym_traffic import workload as wl
    from gym_traffic.envs.roadgraph import GridRoad
    from oracle.oracle import OracleEnv
    total, m, n, L, C = 6, 2, 2, 120.0, 12
    g = GridRoad(m, n, L)
    g.generate_entrypoints(0)
    env = OracleEnv(m, n, L, C, g.dest, g.phases, g.nexts, n_envs=total)
    env.reset(np.zeros(env.I, np.int32))
    snaps = []
    for t in range(30):
        ids = np.arange(total)
        obs, rew, done = env.step(wl.cycle_actions(ids, env.I, t, period=5),
                                  [wl.spawn_roads_for_tick(g.entrypoints, t + int(k), period=3) for k in ids])
        if (t + 1) % 10 == 0:
            snaps.append((obs.copy(), rew.copy(), done.copy()))
    return snaps, env.vehicle_updates


def test_shard_range_partitions():
    from gym_traffic.distributed import shard_range
    for total in (1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_gather_equals_single_process(world, tmp_path):
    """2 ranks (3 + 3 envs) and 4 ranks (2 + 2 + 1 + 1: uneven shards, padded to the largest)"""
    port = free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    snaps, updates = _single_process_reference()
    assert float(z["t_max"][0]) == float(world)
    assert float(z["updates"][0]) == float(updates) and updates > 0
    for i, (obs, rew, done) in enumerate(snaps):
        assert np.array_equal(z["s%d_0" % i], obs)       # env-id order, sharding-independent
        assert np.array_equal(z["s%d_1" % i], rew)
        assert np.array_equal(z["s%d_2" % i], done)


def test_single_process_gather_is_a_snapshot():
    from gym_traffic.distributed import RolloutGather
    gth = RolloutGather(3, 5, 2, "cpu")
    obs = torch.arange(15, dtype=torch.int32).reshape(3, 5)
    rew = torch.ones(3, 2)
    done = torch.zeros(3, dtype=torch.uint8)
    gth.start(obs, rew, done)
    obs += 100                                            # live buffer moves on; the snapshot must not
    o, r, d = gth.result()
    assert o[0, 0].item() == 0 and torch.equal(r, torch.ones(3, 2)) and d.sum().item() == 0


def _uneven_worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gym_traffic.distributed import shard_range, RolloutGather
        lo, hi = shard_range(7, rank, world)                       # 4 + 3 envs
        n, L, I = hi - lo, 6, 2
        sizes = [b - a for a, b in (shard_range(7, r, world) for r in range(world))]
        g = RolloutGather(n, L, I, "cpu", counts=sizes)
        ids = torch.arange(lo, hi, dtype=torch.int32)
        for step in range(5):                                      # starts never wait for the previous one
            g.start(ids[:, None].repeat(1, L) + 100 * step, (ids[:, None].repeat(1, I) + 0.25).float(),
                    (ids % 2).to(torch.uint8))
            assert sum(w is not None for w in g.pending) <= g.DEPTH
        res = g.result()
        if rank == 0:
            obs, rew, done = res
            assert g.collectives == 5                              # ONE gather per snapshot
            np.savez(os.path.join(out_dir, "uneven.npz"), obs=obs.numpy(), rew=rew.numpy(), done=done.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_uneven_shards_one_collective_per_snapshot(world, tmp_path):
    mp.spawn(_uneven_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "uneven.npz"))
    ids = np.arange(7)
    assert np.array_equal(z["obs"], np.repeat(ids[:, None], 6, 1) + 400)
    assert np.array_equal(z["rew"], np.repeat(ids[:, None], 2, 1) + 0.25)
    assert np.array_equal(z["done"], ids % 2)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE: the parent starts the ranks (torchrun, 127.0.0.1),
    relays exactly one JSON line and returns the job's exit code - here with the CPU-only plumbing
    check in place of the GPU step."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"]
    ok = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                        timeout=240)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [ln for ln in ok.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] == 1.0 and out["t_max"] == 2.0
    bad = subprocess.run(cmd + ["--selftest-fail-rank", "1"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=240)
    assert bad.returncode != 0 and bad.stdout.strip() == ""        # a failed rank: no line, non-zero exit


def test_bench_refuses_more_ranks_than_gpus():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "TFX_BENCH_REHEARSAL")}
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=120)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
