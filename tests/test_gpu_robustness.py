"""Robustness of the boundary on the GPU: concurrent handles from host threads (the reference's A3C
steps one env per thread, a3c.py:69-72), flags that change between steps (the reference re-reads
FLAGS every tick), C-ABI argument errors, and streams other than the default one."""
import threading

import numpy as np
import pytest

from oracle.oracle import OracleEnv

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic.core import TfxEngine  # noqa: E402
from gym_traffic import _native as nat  # noqa: E402
from gym_traffic import workload as wl  # noqa: E402


def run_engine(seed, out, T=150):
    rng = np.random.RandomState(seed)
    eng = TfxEngine(3, 3, 150.0, 16, n_envs=2, planes=2)
    orc = OracleEnv(3, 3, 150.0, 16, eng.dest, eng.phases, eng.nexts, n_envs=2)
    ph = rng.randint(2, size=(2, eng.I)).astype(np.int32)
    eng.reset(ph)
    orc.reset(ph)
    ok = True
    for t in range(T):
        act = rng.randint(2, size=(2, eng.I)).astype(np.int32)
        roads = [rng.choice(eng.entrypoints, size=rng.randint(0, 3)).tolist() for _ in range(2)]
        cnt = np.zeros((2, eng.n_entry), np.int32)
        for k, rl in enumerate(roads):
            for rd in rl:
                cnt[k, eng.entry_index[rd]] += 1
        eng.set_spawns(counts=cnt)
        eng.set_actions(act)
        eng.step(1)
        orc.step(act, roads)
        if t % 25 == 24:
            ok &= np.array_equal(eng.leading.cpu().numpy(), orc.leading)
            ok &= np.array_equal(eng.obs.cpu().numpy(), orc.obs)
    out[seed] = bool(ok)


def test_concurrent_handles_from_threads():
    out = {}
    threads = [threading.Thread(target=run_engine, args=(s, out)) for s in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert out == {0: True, 1: True, 2: True, 3: True}


def test_non_default_stream():
    a = wl.setup_engine("cfg1", envs=8)
    b = wl.setup_engine("cfg1", envs=8)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        a.step(30)
    s.synchronize()
    b.step(30)
    torch.cuda.synchronize()
    assert torch.equal(a.leading, b.leading) and torch.equal(a.obs, b.obs)


def test_flag_change_between_steps_rebuilds_engine():
    import gym_traffic  # noqa: F401
    import gym
    from gym_traffic.envs.roadgraph import GridRoad
    from gym_traffic.flags import update_flags
    try:
        env = gym.make('traffic-v0')
        env.set_graph(GridRoad(2, 2, 120), capacity=12)
        env.seed_generator(3)
        env.reset_entrypoints()
        np.random.seed(3)
        env.reset()
        for _ in range(20):
            env.step(env.action_space.sample())
        before = (np.asarray(env.leading).copy(), env.obs.copy(), float(env.steps))
        update_flags(learn_switch=True)                 # picked up on the next step, state kept
        env.step(np.zeros(4, np.int32))                 # learn_switch: action 0 = keep the phase
        assert np.array_equal(env.current_phase, before[1][-8:-4])
        assert float(env.steps) == before[2] + 1
        assert env.engine.cfg.learn_switch == 1
    finally:
        update_flags(learn_switch=False)


def test_cabi_argument_errors_on_device():
    import ctypes as C
    eng = TfxEngine(2, 2, 100.0, 10, n_envs=1)
    lib = eng.lib
    assert lib.tfx_step(eng.h, -1, None) == -1
    assert lib.tfx_set_actions(eng.h, 99, None, 0, 0) == -1
    assert lib.tfx_set_spawns(eng.h, nat.SPAWN_COUNTS, None, 0, 0) == -1
    assert lib.tfx_cars_on_roads(eng.h, None, None) == -1
    assert b"null" in lib.tfx_last_error()
    cfg = nat.TfxConfig()
    C.memmove(C.byref(cfg), C.byref(eng.cfg), C.sizeof(cfg))
    cfg.car_delta = 3.0
    h = C.c_void_p()
    assert lib.tfx_create(C.byref(cfg), C.byref(h)) == -1 and b"delta" in lib.tfx_last_error()
    cfg.car_delta = 4.0
    cfg.capacity = 300
    assert lib.tfx_create(C.byref(cfg), C.byref(h)) == -1
    h2 = C.c_void_p()
    cfg.capacity = 10
    assert lib.tfx_create(C.byref(cfg), C.byref(h2)) == 0
    assert lib.tfx_step(h2, 1, None) == -2                # buffers not bound yet
    assert lib.tfx_destroy(h2) == 0
