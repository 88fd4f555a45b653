"""The N > 1 path on ONE GPU: two ranks share device 0 and talk over gloo (the rehearsal mode of
bench.py) - the env-sharded step with global env ids, the packed (obs | reward | done) gather to rank 0
and the self-launching benchmark.  RCCL itself needs more than one GPU: test_two_gpus_rccl_gather runs
the real `nccl` RolloutGather as soon as two devices are visible and is skipped on a one-GPU box."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gym_traffic import workload as wl
        from gym_traffic.distributed import shard_range, RolloutGather
        lo, hi = shard_range(7, rank, world)                       # 4 + 3 envs
        eng = wl.setup_engine("cfg1", device="cuda:0", envs=hi - lo, env_id_offset=lo)
        sizes = [b - a for a, b in (shard_range(7, r, world) for r in range(world))]
        gather = RolloutGather(hi - lo, eng.obs_len, eng.I, eng.device, counts=sizes)
        for _ in range(3):
            eng.step(10)
            gather.start(eng.obs, eng.rewards, eng.done)
        res = gather.result()
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), obs=res[0].cpu().numpy(), rew=res[1].cpu().numpy(),
                     done=res[2].cpu().numpy(), collectives=gather.collectives)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    from gym_traffic import workload as wl
    mp.spawn(_rank, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    ref = wl.setup_engine("cfg1", envs=7)
    ref.step(30)
    assert int(z["collectives"]) == 3
    assert np.array_equal(z["obs"], ref.obs.cpu().numpy())          # env-id order, sharding-independent
    assert np.array_equal(z["rew"], ref.rewards.cpu().numpy())
    assert z["done"].shape == (7,)


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_launches_its_ranks_rehearsal(ranks):
    """`bench.py --gpus N` as the driver starts it, N ranks on GPU 0 over gloo: one JSON line, and in it what a scaling
    curve needs beside the aggregate - the ranks' own times, the gather's duration, the spread over the timed regions."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TFX_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "20", "--warmup", "5",
                        "--envs", "256", "--repeats", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "weak" and out["value"] > 0
    assert "gather" in out["config"]["parallelism"] and out["config"]["envs_per_gpu"] == 256
    assert out["repeats"] == 3 and out["spread"][0] <= out["value"] <= out["spread"][1]
    assert out["per_rank_ms_per_step"]["min"] <= out["per_rank_ms_per_step"]["max"] == out["ms_per_step"]
    assert out["t_max_over_t_min"] >= 1.0 and out["gather_ms_per_snapshot"] > 0
    assert out["rccl_ranks_seen"] == 0          # (a rehearsal: gloo, no RCCL communicator)
    assert out["agent_decision_ms"] > 0 and out["config"]["settle_ticks"] >= 0


def test_snapshot_is_not_torn_by_the_ticks_that_follow():
    """start() returns at once and the caller steps on: the snapshot must hold the tick it was taken at, not
    whatever the following ticks wrote into obs / rewards / done while the copy was in flight."""
    from gym_traffic import workload as wl
    from gym_traffic.distributed import RolloutGather
    eng = wl.setup_engine("cfg2", envs=512)
    gather = RolloutGather(eng.E, eng.obs_len, eng.I, eng.device)
    for rep in range(6):
        eng.step(10)
        want = (eng.obs.clone(), eng.rewards.clone(), eng.done.clone())
        gather.start(eng.obs, eng.rewards, eng.done)
        eng.step(7)                       # overwrites all three buffers right behind the snapshot
        got = gather.result()
        for a, b in zip(got, want):
            assert torch.equal(a, b), rep
    assert gather.collectives == 0        # a single rank gathers nothing


def _rccl_rank(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    try:
        from gym_traffic import workload as wl
        from gym_traffic.distributed import shard_range, RolloutGather
        lo, hi = shard_range(301, rank, world)
        eng = wl.setup_engine("cfg2", device=device, envs=hi - lo, env_id_offset=lo)
        sizes = [b - a for a, b in (shard_range(301, r, world) for r in range(world))]
        gather = RolloutGather(hi - lo, eng.obs_len, eng.I, device, counts=sizes)
        assert gather.stage_dev == device                      # RCCL: the snapshots stay on the device
        for _ in range(4):
            eng.step(10)
            gather.start(eng.obs, eng.rewards, eng.done)
        eng.step(5)                                                # the gather of tick 40 overlaps these ticks
        res = gather.result()
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), obs=res[0].cpu().numpy(), rew=res[1].cpu().numpy(),
                     done=res[2].cpu().numpy(), collectives=gather.collectives)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _rccl_single_rank(rank, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        from gym_traffic import workload as wl
        from gym_traffic.distributed import RolloutGather
        eng = wl.setup_engine("cfg2", device=device, envs=96)
        gather = RolloutGather(96, eng.obs_len, eng.I, device, single_rank_collective=True)
        assert gather.stage_dev == device and gather.collect
        snaps = []
        for _ in range(5):
            eng.step(10)
            snaps.append([t.clone() for t in (eng.obs, eng.rewards, eng.done)])
            gather.start(eng.obs, eng.rewards, eng.done)
            eng.step(3)                                            # ticks run on while the gather is in flight
            got = gather.result()
            for a, b in zip(got, snaps[-1]):
                assert torch.equal(a, b)
        t = torch.tensor([3.5], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        np.savez(os.path.join(out_dir, "one.npz"), collectives=gather.collectives, t=t.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_rccl_process_group_of_one_rank(tmp_path):
    """The `nccl` branch of RolloutGather and of bench.py on a one-GPU box: an RCCL communicator of one rank, the
    gather issued on the side stream through RCCL for real (single_rank_collective), work handles, barrier,
    float64 reductions - everything but a second device."""
    mp.spawn(_rccl_single_rank, args=(free_port(), str(tmp_path)), nprocs=1, join=True)
    z = np.load(os.path.join(str(tmp_path), "one.npz"))
    assert int(z["collectives"]) == 5 and float(z["t"][0]) == 3.5


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs two GPUs (this box has one)")
def test_rccl_gather_on_every_visible_gpu_equals_one_process(tmp_path):
    """cfg3's code path for real: one rank per GPU - as many as the box shows, up to 8 - backend nccl (= RCCL over
    xGMI), uneven shards (301 envs never divide evenly over 2..8 ranks, so the padding of the smaller shards is
    exercised), four snapshots with ticks running behind each.  Runs the first time a multi-GPU box appears."""
    from gym_traffic import workload as wl
    world = min(8, torch.cuda.device_count())
    mp.spawn(_rccl_rank, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    ref = wl.setup_engine("cfg2", envs=301)
    for _ in range(4):
        ref.step(10)
    assert int(z["collectives"]) == 4
    assert np.array_equal(z["obs"], ref.obs.cpu().numpy())
    assert np.array_equal(z["rew"], ref.rewards.cpu().numpy())
    assert np.array_equal(z["done"], ref.done.cpu().numpy())
