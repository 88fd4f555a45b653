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


def test_fuzz_small_configurations_vs_oracle():
    """Thirty random small configurations - grid shape, capacity, road length, rate, learn_switch,
    entry sides, batch size, layout, arrival density - each stepped free-running for 40 ticks in
    uneven multi-tick calls with per-tick buffers: bit-equal to the oracle at the end of every call."""
    from gym_traffic.core import TfxEngine
    from oracle.oracle import OracleEnv
    from test_gpu_parity import assert_same_state, counts
    rng = np.random.RandomState(20251003)
    for trial in range(30):
        m, n = int(rng.randint(1, 5)), int(rng.randint(1, 5))
        C = int(rng.choice([3, 4, 6, 9, 10, 17, 20, 33, 34, 66]))
        L = float(rng.choice([30.0, 75.0, 140.0, 250.0]))
        rate = float(rng.choice([0.25, 0.5, 1.0]))
        ls = bool(rng.randint(2))
        spec = int(rng.choice([0, 0, 0b0001, 0b1010, 0b1110]))
        E = int(rng.choice([1, 2, 5, 70]))
        layout = str(rng.choice(["ring", "transposed"]))
        planes = 2 if layout == "transposed" else int(rng.choice([2, 3]))
        eng = TfxEngine(m, n, L, C, n_envs=E, rate=rate, learn_switch=ls, entry_spec=spec, planes=planes,
                        layout=layout)
        orc = OracleEnv(m, n, L, C, eng.dest, eng.phases, eng.nexts, n_envs=E, rate=rate, learn_switch=ls)
        ph = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        eng.reset(ph)
        orc.reset(ph)
        dens = rng.choice([0.1, 0.5, 1.5])
        t = 0
        while t < 40:
            k = int(min(40 - t, rng.choice([1, 1, 2, 3, 7])))
            acts = rng.randint(2, size=(k, E, eng.I)).astype(np.int32)
            roads = [[(rng.choice(eng.entrypoints, size=rng.poisson(dens)).tolist() if eng.n_entry else [])
                      for _ in range(E)] for _ in range(k)]
            eng.set_actions(acts, per_tick=True)
            eng.set_spawns(counts=np.stack([counts(eng, r) for r in roads]), per_tick=True)
            eng.step(k)
            done = np.zeros(E, bool)
            for j in range(k):
                done |= orc.step(acts[j], roads[j])[2].astype(bool)
            assert np.array_equal(eng.done.cpu().numpy().astype(bool), done), (trial, t)
            t += k
            assert_same_state(eng, orc, "trial %d (%dx%d C=%d E=%d %s) tick %d" % (trial, m, n, C, E, layout, t))


def test_refresh_refuses_a_stale_staging_copy():
    """An edit through a reference to eng.xv kept across a step must not be dropped silently (nor pushed over the
    live cars): refresh() raises, a fresh access + edit + refresh() lands."""
    from gym_traffic import workload as wl
    from gym_traffic._native import TfxError
    eng = wl.setup_engine("cfg1", envs=4)
    xv = eng.xv                      # staging copy, fresh
    eng.step(3)                      # the cars move: `xv` is stale now
    xv[0, 0, 2, 0] = 123.0
    with pytest.raises(TfxError):
        eng.refresh()
    eng.refresh(cars=False)          # explicit: tails only, the cars stay as they are
    fresh = eng.xv
    assert float(fresh[0, 0, 2, 0]) != 123.0
    ld = int(eng.leading[0, 5])
    slot = ld + 1 if ld + 1 < eng.C else 1
    fresh[0, 5, slot, 1] = 0.25      # the head car of road 5 slows down
    eng.refresh()
    assert float(eng.xv[0, 5, slot, 1]) == 0.25


@pytest.mark.parametrize("call", ["step", "agent_step"])
@pytest.mark.parametrize("path", ["split", "one_range"])
def test_failed_launch_in_the_middle_of_a_sequence_leaves_a_usable_handle(call, path, monkeypatch):
    """tfx_debug_fail_after makes the n-th launch of a call fail: the call returns TFX_EDEVICE with a message, and the
    handle is as usable as before - the second stream joined back, agent mode / reward accumulation / the half being
    enqueued restored (round-3 finding: early returns skipped all of that).  After reloading the state the same handle
    reproduces an untouched handle's run bit for bit, in both kinds of call."""
    monkeypatch.setenv("TFX_RESIDENT", "0")
    monkeypatch.setenv("TFX_PAIRS", "2")
    monkeypatch.setenv("TFX_TAIL", "2")
    monkeypatch.setenv("TFX_SPLIT", "2" if path == "split" else "0")
    a = wl.setup_engine("cfg1", envs=6)
    b = wl.setup_engine("cfg1", envs=6)
    snap = [t.clone() for t in (a.xv, a.leading, a.lastcar, a.obs, a.rewards, a.waiting, a.passed_dst)]

    def restore(e):
        e.reset(np.zeros((1, e.I), np.int32))
        ring = e.xv
        ring.copy_(snap[0])
        for dst, src in zip((e.leading, e.lastcar, e.obs, e.rewards, e.waiting, e.passed_dst), snap[1:]):
            dst.copy_(src)
        e.refresh()

    for n_fail in (1, 2, 3, 5, 8):
        nat.check(a.lib.tfx_debug_fail_after(a.h, n_fail))
        with pytest.raises(nat.TfxError, match="injected launch failure"):
            if call == "step":
                a.step(10)
            else:
                a.agent_step(10, remi=True)
        torch.cuda.synchronize()
        restore(a)
        restore(b)
        # plain ticks AND a fused decision on the handle that failed == the same on the untouched one
        a.step(7)
        b.step(7)
        oa = [t.clone() for t in a.agent_step(6, remi=False)]
        ob = [t.clone() for t in b.agent_step(6, remi=False)]
        for x, y in zip(oa, ob):
            assert torch.equal(x, y), (n_fail,)
        for name in ("leading", "lastcar", "obs", "rewards", "waiting", "passed_dst"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (n_fail, name)
        assert torch.equal(a.xv, b.xv), n_fail
    nat.check(a.lib.tfx_debug_fail_after(a.h, 0))
