"""The agent-facing wrapper stack over the GPU env against the SAME stack of the reference over the
reference env (tests/golden/wrappers/stack_*.npz, captured by oracle/gen_golden_wrappers.py):
Repeater -> Warmup -> Remi -> Localize -> Squish -> History assembled by make_env from the same
flags and seeds.  Both Repeater paths are checked: the fused device decision (tfx_agent_step via
TrafficEnv.repeat) and the plain tick loop - including decisions cut short by an overflow, after
which the arrival generator and the clock must stand where the reference's loop left them."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import gym_traffic  # noqa: E402,F401
from gym_traffic.flags import update_flags  # noqa: E402
from gym_traffic.wrappers import agent as A  # noqa: E402
from gym_traffic.wrappers import vec as V  # noqa: E402
from gym_traffic.envs.vec_env import TrafficVecEnv  # noqa: E402

STACKS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "wrappers", "stack_*.npz")))
DEFAULTS = dict(poisson=True, rate=0.5, local_cars_per_sec=0.12, entry='all', learn_switch=False, mode='train',
                light_secs=5, warmup_lights=0, remi=True, local_weight=1, squish_rewards=False, history=1,
                render=False, light_iterations=None)


def build(sc, fused):
    update_flags(poisson=bool(sc["poisson"]), rate=0.5, local_cars_per_sec=float(sc["lcps"]), entry='all',
                 learn_switch=False, mode=sc["mode"], light_secs=sc["light_secs"],
                 warmup_lights=sc["warmup_lights"], remi=sc["remi"], local_weight=sc["local_weight"],
                 squish_rewards=sc["squish_rewards"], history=sc["history"], render=False,
                 light_iterations=None)
    env = A.make_env(sc["m"], sc["n"], sc["L"], seed=sc["seed"], capacity=sc["C"])
    rep = env
    while not isinstance(rep, A.RepeaterBase):
        rep = rep.env
    rep.fused = fused
    return env


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", STACKS)
def test_stack_reproduces_reference_stack(name, fused):
    z = np.load(os.path.join(GOLDEN_DIR, "wrappers", name + ".npz"))
    sc = json.loads(str(z["scenario"]))
    try:
        env = build(sc, fused)
        base = env.unwrapped
        np.random.seed(sc["seed"])
        first = np.array(env.reset())
        assert first.shape == z["reset"].shape and first.dtype == z["reset"].dtype
        assert np.array_equal(first, z["reset"])
        lt = []
        for k in range(sc["decisions"]):
            obs, rew, done, info = env.step(z["actions"][k])
            assert np.array_equal(np.array(obs), z["obs"][k]), (name, k)
            assert np.array_equal(np.atleast_1d(np.asarray(rew, np.float64)), z["rewards"][k]), (name, k)
            assert bool(done) == bool(z["done"][k]), (name, k)
            assert float(base.steps) == z["env_steps"][k] and base.generated_cars == z["generated_cars"][k]
            if info:
                lt.extend(np.asarray(info['light_times'], np.float64).tolist())
            assert len(lt) == z["light_off"][k + 1]
        assert np.array_equal(np.asarray(lt), z["light_times"])
        assert (getattr(base.engine, "_aobs", None) is not None) == fused     # the path under test ran
        assert np.array_equal(np.asarray(base.trip_times, np.float64), z["trip_times"])
        assert int(np.sum(base.cars_on_roads())) == int(z["unfinished"])
        assert np.array_equal(np.array(base.leading), z["final_leading"])
        assert np.array_equal(np.array(base.lastcar), z["final_lastcar"])
    finally:
        update_flags(**DEFAULTS)


def test_fused_and_looped_repeater_leave_the_same_env():
    """Beyond the outputs: after a decision cut short by an overflow both paths agree on every car."""
    z = np.load(os.path.join(GOLDEN_DIR, "wrappers", "stack_localize_overflow.npz"))
    sc = json.loads(str(z["scenario"]))
    try:
        envs = []
        for fused in (True, False):
            env = build(sc, fused)
            np.random.seed(sc["seed"])
            env.reset()
            for k in range(sc["decisions"]):
                env.step(z["actions"][k])
            envs.append(env.unwrapped)
        a, b = envs
        assert a.engine.tick == b.engine.tick == int(z["env_steps"][-1])
        assert np.array_equal(np.array(a.leading), np.array(b.leading))
        sa, sb = a.state.numpy(), b.state.numpy()
        from oracle.oracle import live_mask
        live = live_mask(np.array(a.leading), np.array(a.lastcar), a.capacity)
        for p in range(3):
            assert np.array_equal(sa[:, p, :][live], sb[:, p, :][live])
        assert a.rand.get_state()[1].tolist() == b.rand.get_state()[1].tolist()
    finally:
        update_flags(**DEFAULTS)


def test_batched_stack_on_device_matches_single_env_stack():
    """VecHistory(VecRemiRepeater(TrafficVecEnv)) env by env against History(Remi(Repeater(TrafficEnv)))
    with the same seeds: E envs, one device submission per decision."""
    E, H, n_dec = 3, 2, 6
    try:
        update_flags(**dict(DEFAULTS, history=H))
        venv = TrafficVecEnv(E, 2, 2, 250.0, capacity=14, spawn='poisson', seed=11, local_cars_per_sec=0.2)
        stack = V.VecHistory(V.VecRemiRepeater(venv, 10, remi=True), H)
        rng = np.random.RandomState(5)
        acts = rng.randint(2, size=(n_dec + H, E, 4)).astype(np.int32)
        ph0 = rng.randint(2, size=(E, 4)).astype(np.int32)
        feed = iter(acts)
        stack.sample_actions = lambda: torch.as_tensor(next(feed)).to(venv.engine.device)
        stack.venv.sample_actions = stack.sample_actions
        got0 = stack.reset(ph0).cpu().numpy().copy()
        outs = []
        for k in range(n_dec):
            o, r, d = stack.step(torch.as_tensor(acts[H + k]).to(venv.engine.device))
            outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
        for e in range(E):
            update_flags(local_cars_per_sec=0.2)
            env = A.make_env(2, 2, 250.0, seed=11 + e, capacity=14)
            base = env.unwrapped
            seq = iter(acts[:, e])
            base.action_space.sample = lambda: ph0[e]            # reset's initial phases
            # the Repeater / History sample their reset actions through the wrapper's action_space
            w = env
            while w is not base:
                w.action_space = base.action_space
                w = w.env
            first = None

            class Feed(object):
                shape, limit, size = base.action_space.shape, base.action_space.limit, base.action_space.size
                calls = 0

                def sample(self):
                    Feed.calls += 1
                    return ph0[e] if Feed.calls == 1 else next(seq)
            sp = Feed()
            w = env
            while True:
                w.action_space = sp
                if w is base:
                    break
                w = w.env
            first = np.array(env.reset())
            assert np.array_equal(first, got0[e]), e
            for k in range(n_dec):
                o, r, d, _ = env.step(acts[H + k, e])
                assert np.array_equal(np.array(o), outs[k][0][e]), (e, k)
                assert np.array_equal(np.asarray(r, np.float32), outs[k][1][e]), (e, k)
                assert bool(d) == bool(outs[k][2][e])
    finally:
        update_flags(**DEFAULTS)


def test_batched_validate_metrics_match_the_reference_stack():
    """Validate-mode metrics in batched form: `light_times` per decision (traffic_test.py:41-46),
    trip times (traffic_env.py:139-157) and `unfinished` (util.py:91-92) from VecRemiRepeater(TrafficVecEnv(validate=
    True)).  Env 0 of the batch is seeded like the reference run captured in stack_validate.npz and must reproduce it
    - observations, Remi rewards, light times, trip times, unfinished cars, final ring indices; the other envs of the
    batch (other seeds) must equal single-env stacks of this package built with those seeds."""
    from gym_traffic.spaces.gspace import GSpace
    z = np.load(os.path.join(GOLDEN_DIR, "wrappers", "stack_validate.npz"))
    sc = json.loads(str(z["scenario"]))
    E, I, n_dec = 3, sc["m"] * sc["n"], sc["decisions"]
    ticks = int(sc["light_secs"] / 0.5)
    try:
        update_flags(**dict(DEFAULTS, mode='validate', local_cars_per_sec=float(sc["lcps"])))
        # the reference run: np.random.seed(seed); reset() draws the initial phases, Repeater._reset one action
        sp = GSpace([I], np.int32(2))
        ph0, a0 = np.zeros((E, I), np.int32), np.zeros((E, I), np.int32)
        for e in range(E):
            np.random.seed(sc["seed"] + e)
            ph0[e], a0[e] = sp.sample(), sp.sample()
        rng = np.random.RandomState(77)
        acts = rng.randint(2, size=(n_dec, E, I)).astype(np.int32)
        acts[:, 0] = z["actions"]
        venv = TrafficVecEnv(E, sc["m"], sc["n"], sc["L"], capacity=sc["C"], spawn='poisson', seed=sc["seed"],
                             local_cars_per_sec=float(sc["lcps"]), validate=True)
        stack = V.VecRemiRepeater(venv, ticks, remi=True)
        stack.sample_actions = lambda: torch.as_tensor(a0).to(venv.engine.device)
        first = stack.reset(ph0).cpu().numpy().copy()
        assert np.array_equal(first[0], z["reset"])
        lt = [[] for _ in range(E)]
        outs = []
        for k in range(n_dec):
            o, r, d = stack.step(torch.as_tensor(acts[k]).to(venv.engine.device))
            t = stack.info['light_times'].cpu().numpy()
            for e in range(E):
                lt[e].extend(t[e][t[e] != 0].astype(np.float64).tolist())
            outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
            assert np.array_equal(outs[-1][0][0], z["obs"][k]), k
            assert np.array_equal(outs[-1][1][0].astype(np.float64), z["rewards"][k]), k
            assert bool(outs[-1][2][0]) == bool(z["done"][k]), k
            assert len(lt[0]) == z["light_off"][k + 1], k
        assert np.array_equal(np.asarray(lt[0]), z["light_times"])
        assert np.array_equal(venv.trip_times(0).astype(np.float64), z["trip_times"])
        unfinished = venv.unfinished().cpu().numpy()
        assert int(unfinished[0]) == int(z["unfinished"])
        assert np.array_equal(venv.engine.leading[0].cpu().numpy(), z["final_leading"])
        assert np.array_equal(venv.engine.lastcar[0].cpu().numpy(), z["final_lastcar"])
        trips = venv.trip_times()
        for e in range(1, E):                       # the other envs: single-env stacks with their seeds
            env = A.make_env(sc["m"], sc["n"], sc["L"], seed=sc["seed"] + e, capacity=sc["C"])
            base = env.unwrapped
            np.random.seed(sc["seed"] + e)
            assert np.array_equal(np.array(env.reset()), first[e]), e
            lte = []
            for k in range(n_dec):
                o, r, d, info = env.step(acts[k, e])
                lte.extend(np.asarray(info['light_times'], np.float64).tolist())
                assert np.array_equal(np.array(o), outs[k][0][e]), (e, k)
                assert np.array_equal(np.asarray(r, np.float32), outs[k][1][e]), (e, k)
            assert lte == lt[e], e
            assert np.array_equal(np.asarray(base.trip_times, np.float32), trips[e]), e
            assert int(np.sum(base.cars_on_roads())) == int(unfinished[e]), e
    finally:
        update_flags(**DEFAULTS)
