"""The N > 1 path on ONE GPU: two ranks share device 0 and talk over gloo (the rehearsal mode of
bench.py) - the env-sharded step with global env ids, the packed (obs | reward | done) gather to rank 0
and the self-launching benchmark.  RCCL itself needs more than one GPU and is exercised by the
driver's multi-GPU runs only."""
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


def test_bench_launches_two_ranks_rehearsal():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TFX_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--envs", "256"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "gather" in out["config"]["parallelism"] and out["config"]["envs_per_gpu"] == 256
