"""The reference's object surface on the GPU: gym.make('traffic-v0') -> TrafficEnv driven exactly
like traffic_test.py / algorithms/*.py drive it, compared with the golden runs captured from the
reference under the same seeds ("identical seeds/spawns"), and TrafficVecEnv against the oracle."""
import numpy as np
import pytest

from conftest import golden_names
from oracle.oracle import OracleEnv, live_mask

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import gym_traffic  # noqa: E402,F401
import gym  # noqa: E402
from gym_traffic.envs.roadgraph import GridRoad  # noqa: E402
from gym_traffic.flags import update_flags  # noqa: E402


def make_env(sc):
    update_flags(poisson=bool(sc["poisson"]), rate=float(sc["rate"]), local_cars_per_sec=float(sc["lcps"]),
                 entry=sc["entry"], learn_switch=bool(sc["learn_switch"]), mode=sc["mode"])
    env = gym.make('traffic-v0')
    env.set_graph(GridRoad(sc["m"], sc["n"], sc["L"]), capacity=sc["C"])
    env.seed_generator(sc["seed"])
    env.reset_entrypoints()
    return env


@pytest.mark.parametrize("name", ["g2x2_s0_poi_c10", "g2x2_s1_reg_c20", "g2x2_entry_one", "g2x2_learnswitch",
                                  "g3x3_default", "g3x2_rect", "g2x2_validate"])
def test_gym_surface_reproduces_reference_run(name, golden_cache):
    g = golden_cache(name)
    sc = g.sc
    try:
        env = make_env(sc)
        assert env.action_space.shape == [sc["m"] * sc["n"]] and env.reward_size == sc["m"] * sc["n"]
        assert np.array_equal(env.graph.entrypoints, g["entrypoints"])
        np.random.seed(sc["seed"])
        obs = env.reset()
        assert obs is env.obs
        assert np.array_equal(env.current_phase, g["init_phase"])
        ri = 0
        for t in range(120):
            # the reference accepts bool / float / int actions alike (a3c.py:60, const0.py:8)
            a = g["actions"][t]
            a = a.astype(bool) if t % 3 == 0 else (a.astype(np.float64) if t % 3 == 1 else a)
            obs, rew, done, info = env.step(a)
            k = t + 1
            assert obs is env.obs and rew is env.rewards and info is None
            assert np.array_equal(obs, g["obs"][k]), (name, k)
            assert np.array_equal(rew, g["rewards"][k]), (name, k)
            assert bool(done) == bool(g["done"][k]), (name, k)
            assert np.array_equal(np.asarray(env.leading), g["leading"][k])
            assert np.array_equal(np.asarray(env.lastcar), g["lastcar"][k])
            assert np.array_equal(np.asarray(env.waiting), g["waiting"][k])
            assert np.array_equal(np.asarray(env.passed_dst).astype(np.uint8), g["passed_dst"][k])
            if k % 10 == 0:
                assert np.array_equal(env.cars_on_roads(), g["cars_on_roads"][ri])
                r = env.unwrapped.remi_reward()
                env.unwrapped.passed_dst[:] = False            # what the Remi wrapper does (traffic_test.py:63)
                assert np.array_equal(r, g["remi_rewards"][ri])
                ri += 1
        assert env.generated_cars == int(g["spawn_off"][120])
        assert float(env.steps) == 120.0
        if sc["mode"] == "validate":
            n = int(g["trip_count"][120])
            assert len(env.trip_times) == n
            assert np.array_equal(np.asarray(env.trip_times, np.float64), g["trip_times"][:n])
        # env.state in the reference's shape (traffic_env.py:364): [R, 10, C], wi / li at the reference's indices,
        # a car's seven constants on its slot, zeros on the fake leader's
        from gym_traffic.envs import traffic_env as te
        st = env.state.numpy()
        assert st.shape == (env.graph.roads, 10, sc["C"]) and env.state.shape == st.shape
        live = live_mask(np.asarray(env.leading), np.asarray(env.lastcar), sc["C"])
        assert live.any()
        for col, val in ((te.li, 4), (te.ai, 3), (te.deltai, 4), (te.v0i, 13.89), (te.bi, 6), (te.ti, 2), (te.s0i, 1)):
            assert np.all(st[:, col, :][live] == np.float32(val)) and np.all(st[:, col, :][~live] == 0)
        rows = np.arange(env.graph.roads)
        assert np.all(st[rows, te.vi, np.asarray(env.leading)] == 0)
        if "state_x" in g and sc["state_every"] == 1:
            # (free-running for 120 ticks: within the drift the float contract allows)
            assert np.allclose(st[:, te.xi, :][live], g["state_x"][120][live], rtol=1e-4, atol=1e-3)
            assert np.array_equal(st[:, te.wi, :][live], g["state_w"][120][live])
    finally:
        update_flags(poisson=True, rate=0.5, local_cars_per_sec=0.12, entry='all', learn_switch=False, mode='train')


def test_wrapper_chain_like_traffic_test():
    """Repeater(10) + Remi as in traffic_test.py:27-64, written against the env's public surface."""
    class Repeater(gym.Wrapper):
        def _step(self, action):
            tot = 0
            for _ in range(10):
                obs, r, done, _ = self.env.step(action)
                tot = tot + r
                if done:
                    break
            return obs.copy(), tot, done, None

    class Remi(gym.Wrapper):
        def _step(self, action):
            obs, _, done, info = self.env.step(action)
            r = self.unwrapped.remi_reward()
            self.unwrapped.passed_dst[:] = False
            return obs, r.copy(), done, info

    update_flags(poisson=True, local_cars_per_sec=0.3)
    try:
        env = gym.make('traffic-v0')
        env.set_graph(GridRoad(3, 3, 250))
        env.seed_generator(5)
        env.reset_entrypoints()
        w = Remi(Repeater(env))
        assert w.reward_size == 9 and w.unwrapped is env
        w.reset()
        tot_cars = 0
        for i in range(12):
            obs, r, done, _ = w.step(w.action_space.sample())
            assert r.shape == (9,) and set(np.unique(r)).issubset({-2, -1.5, -1, -.5, 0, .5, 1, 1.5, 2})
            tot_cars = int(env.cars_on_roads().sum())
        # Repeater stops a repeat early when the env reports overflow (done), like the reference's
        assert tot_cars > 20 and 60.0 <= float(env.steps) <= 120.0
        frame = env.render(mode='rgb_array')
        assert frame.ndim == 3 and frame.shape[2] == 3
        env.render(close=True)
    finally:
        update_flags(local_cars_per_sec=0.12)


def test_vec_env_seeded_poisson_vs_oracle():
    """TrafficVecEnv: E envs with per-env seeded generators == E oracle envs fed the same schedules;
    an env's trajectory does not depend on which shard it sits in (env_id_offset)."""
    from gym_traffic.envs.vec_env import TrafficVecEnv
    from gym_traffic.spawner import SpawnSchedule
    E, T = 6, 80
    kw = dict(m=3, n=3, length=180.0, capacity=16, local_cars_per_sec=0.3, spawn='poisson', seed=11)
    vec = TrafficVecEnv(E, **kw)
    shard = TrafficVecEnv(2, env_id_offset=3, **kw)              # global envs 3, 4
    eng = vec.engine
    orc = OracleEnv(3, 3, 180.0, 16, eng.dest, eng.phases, eng.nexts, n_envs=E)
    sched = [SpawnSchedule(np.random.RandomState(11 + k), True, eng.entrypoints, lambda: (0.3 * 3 * 4, 0.5))
             for k in range(E)]
    ph = np.random.RandomState(0).randint(2, size=(E, eng.I)).astype(np.int32)
    vec.reset(ph)
    shard.reset(ph[3:5])
    orc.reset(ph)
    rng = np.random.RandomState(1)
    for t in range(T):
        act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        obs, rew, done = vec.step(torch.as_tensor(act))
        shard.step(torch.as_tensor(act[3:5]))
        oo, orw, od = orc.step(act, [s.next_tick() for s in sched])
        assert np.array_equal(obs.cpu().numpy(), oo) and np.array_equal(rew.cpu().numpy(), orw)
        assert np.array_equal(done.cpu().numpy(), od)
    assert torch.equal(shard.engine.obs, vec.engine.obs[3:5])
    assert torch.equal(shard.engine.leading, vec.engine.leading[3:5])
    x, v, _ = eng.planes_numpy()
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    for k in range(E):
        live = live_mask(ld[k], lc[k], 16)
        assert np.array_equal(x[k][live], orc.x[k][live]) and np.array_equal(v[k][live], orc.v[k][live])
    assert int(eng.cars_on_roads_flat().sum()) > 30


def test_reset_envs_resets_only_the_selected_envs():
    """tfx_reset_envs: the episode boundary of a batched rollout.  Env by env against single-env
    oracles: the selected envs restart from an empty network with new phases, the others continue
    bit for bit; the batch clock keeps running."""
    from gym_traffic.core import TfxEngine
    E, m, n, L, C = 6, 2, 3, 90.0, 8
    eng = TfxEngine(m, n, L, C, n_envs=E, planes=2)
    orcs = [OracleEnv(m, n, L, C, eng.dest, eng.phases, eng.nexts) for _ in range(E)]
    rng = np.random.RandomState(3)
    ph = rng.randint(2, size=(E, eng.I)).astype(np.int32)
    eng.reset(ph)
    for k, o in enumerate(orcs):
        o.reset(ph[k])
    resets = 0
    for t in range(80):
        act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        roads = [rng.choice(eng.entrypoints, size=rng.randint(0, 3)).tolist() for _ in range(E)]
        c = np.zeros((E, eng.n_entry), np.int32)
        for k, rl in enumerate(roads):
            for rd in rl:
                c[k, eng.entry_index[int(rd)]] += 1
        eng.set_spawns(counts=c)
        eng.set_actions(act)
        eng.step(1)
        done = eng.done.cpu().numpy().astype(bool)
        for k, o in enumerate(orcs):
            o.steps[:] = t
            _, _, d = o.step(act[k], [roads[k]])
            assert bool(d[0]) == bool(done[k]), (t, k)
        if t % 9 == 8:
            mask = done | (rng.rand(E) < 0.3)
            newph = rng.randint(2, size=(E, eng.I)).astype(np.int32)
            eng.reset_envs(mask, newph)
            for k in np.nonzero(mask)[0]:
                orcs[k].reset(newph[k])
                orcs[k].rewards[:] = eng.rewards[k].cpu().numpy()      # _reset leaves rewards / detected stale
                orcs[k].obs[0, eng.r:2 * eng.r] = eng.obs[k, eng.r:2 * eng.r].cpu().numpy()
            resets += int(mask.sum())
            assert int(eng.done[torch.as_tensor(mask)].sum()) == 0
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        xv = eng.xv.cpu().numpy()
        for k, o in enumerate(orcs):
            assert np.array_equal(ld[k], o.leading[0]) and np.array_equal(lc[k], o.lastcar[0]), (t, k)
            assert np.array_equal(eng.obs[k].cpu().numpy(), o.obs[0]), (t, k)
            live = live_mask(ld[k], lc[k], C)
            assert np.array_equal(xv[k][live][:, 0].view(np.int32), o.x[0][live].view(np.int32)), (t, k)
    assert resets > 10 and eng.tick == 80
