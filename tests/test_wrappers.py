"""The wrapper stack on CPU: this repository's History / Strobe / Last /
Warmup wrappers against outputs captured from the REFERENCE's wrapper classes over the same
deterministic CounterEnv (tests/golden/wrappers/wrappers_counter.npz, oracle/gen_golden_wrappers.py),
plus the parts no golden covers: the GSpace adapters, the reward wrappers, the scalar-limit Strobe
and the batched (torch) wrappers against the single-env ones."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

import gym_traffic  # noqa: F401  (installs the gym compat layer when gym is absent)
import gym
from gym_traffic.spaces.gspace import GSpace
from gym_traffic.wrappers.history import HistoryWrapper
from gym_traffic.wrappers.strobe import StrobeWrapper, LastWrapper
from gym_traffic.wrappers.warmup import WarmupWrapper
from gym_traffic.wrappers.gspace import GSpaceWrapper, UnGSpaceWrapper
from gym_traffic.wrappers.agent import Repeater, Remi, LocalizeWrapper, SquishReward
from gym_traffic.flags import update_flags
from oracle.fake_env import make_counter_env

Z = np.load(os.path.join(GOLDEN_DIR, "wrappers", "wrappers_counter.npz"))
CASES = json.loads(str(Z["cases"]))


def build(case):
    env = make_counter_env(gym, GSpace, **case["env"])
    w = case["wrap"]
    if w[0] == "history":
        return HistoryWrapper(w[1])(env)
    if w[0] == "warmup":
        return WarmupWrapper(w[1])(env)
    if w[0] == "last":
        return LastWrapper(w[1])(env)
    if w[0] == "strobe":
        return StrobeWrapper(w[1], w[2], w[3])(env)
    if w[0] == "warmup+history":
        return HistoryWrapper(w[2])(WarmupWrapper(w[1])(env))
    raise KeyError(w[0])


@pytest.mark.parametrize("name", sorted(CASES))
def test_wrapper_matches_reference_wrapper(name):
    case = CASES[name]
    np.random.seed(123)                      # the wrappers sample reset/warm-up actions from the global RNG
    env = build(case)
    got = np.array(env.reset())
    want = Z["%s/reset" % name]
    assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want)
    for k in range(case["steps"]):
        obs, rew, done, _ = env.step(Z["%s/a%d" % (name, k)])
        obs = np.array(obs)
        want = Z["%s/obs%d" % (name, k)]
        assert obs.shape == want.shape and np.array_equal(obs, want), (name, k)
        assert np.array_equal(np.asarray(rew, np.float64), Z["%s/rew%d" % (name, k)]), (name, k)
        assert bool(done) == bool(Z["%s/done%d" % (name, k)])


def test_wrapper_spaces_and_reward_size():
    env = HistoryWrapper(4)(make_counter_env(gym, GSpace))
    assert env.observation_space.shape == [4, 6] and env.observation_space.size == 24
    assert env.reward_size == 3 and env.action_space.shape == [3]
    env = StrobeWrapper(6, 3)(make_counter_env(gym, GSpace))
    assert env.observation_space.shape == [3, 6]
    with pytest.raises(AssertionError):
        StrobeWrapper(7, 3)(make_counter_env(gym, GSpace))
    with pytest.raises(IndexError):          # stepping a History window that was never reset
        HistoryWrapper(2)(make_counter_env(gym, GSpace)).step(np.zeros(3, np.int32))
    with pytest.raises(AssertionError, match="warmup"):
        WarmupWrapper(5)(make_counter_env(gym, GSpace, done_at=3)).reset()


def test_strobe_scalar_limit_equals_array_limit():
    """The reference's StrobeWrapper cannot be constructed over a scalar-limit GSpace (0-d mask);
    here it can, and behaves like the array-limit form."""
    a = StrobeWrapper(6, 3, [0, 2])(make_counter_env(gym, GSpace, array_limit=True))
    b = StrobeWrapper(6, 3, [0, 2])(make_counter_env(gym, GSpace, array_limit=False))
    np.random.seed(5)
    ra = np.array(a.reset())
    np.random.seed(5)
    rb = np.array(b.reset())
    assert np.array_equal(ra, rb)
    for k in range(3):
        act = np.array([k % 2, 1, 0], np.int32)
        oa, wa, da, _ = a.step(act)
        ob, wb, db, _ = b.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(wa, wb) and da == db


class BoxEnv(gym.Env):
    """A classic Box/Discrete env (CartPole-like surface) for the GSpace adapters."""

    def __init__(self):
        self.observation_space = gym.spaces.Box(-np.ones(4, np.float32) * 3, np.ones(4, np.float32) * 3, shape=(4,))
        self.action_space = gym.spaces.Discrete(2)
        self.seen = []

    def _reset(self):
        return [0.0, 1.0, 2.0, 3.0]

    def _step(self, action):
        self.seen.append(action)
        return np.arange(4) + action, 1.5, False, {}


def test_gspace_adapters():
    inner = BoxEnv()
    env = GSpaceWrapper(inner)
    assert env.observation_space.shape == (4,) and env.observation_space.limit == np.float32(3)
    assert env.action_space.shape == [1] and env.action_space.limit == np.int32(2)
    assert np.array_equal(env.reset(), [0, 1, 2, 3])
    obs, rew, done, info = env.step(np.array([1], np.int32))
    assert inner.seen == [1] and isinstance(inner.seen[0], int)
    assert np.array_equal(obs, [1, 2, 3, 4]) and rew.shape == (1,) and rew[0] == 1.5 and not done

    class Rec(gym.Env):
        def __init__(self):
            self.action_space = GSpace([2, 3], np.int32(2))
            self.observation_space = GSpace([5], np.int32(9))
            self.reward_size = 6

        def _reset(self):
            return np.zeros(5, np.int32)

        def _step(self, action):
            self.last = action
            return np.ones(5, np.int32), np.array([1.0, 2.0, 6.0], np.float32), True, None

    rec = Rec()
    flat = UnGSpaceWrapper(rec)
    assert flat.action_space.n == 6 and flat.observation_space.shape == [5]
    obs, rew, done, _ = flat.step(4)
    assert tuple(int(v) for v in rec.last) == (1, 1) and rew == 3.0 and done


class RewardEnv(gym.Env):
    def __init__(self, rewards):
        self.rw = np.asarray(rewards, np.float32)
        self.reward_size = self.rw.size
        self.action_space = GSpace([self.rw.size], np.int32(2))
        self.observation_space = GSpace([2], np.int32(3))

    def _reset(self):
        return np.zeros(2, np.int32)

    def _step(self, action):
        return np.zeros(2, np.int32), self.rw, False, None


def test_reward_wrappers():
    a = np.array([1.0, -2.0, 4.0, 0.5], np.float32)
    try:
        update_flags(local_weight=3)
        got = LocalizeWrapper(RewardEnv(a)).step(None)[1]
        want = np.array([(a.sum() + 2 * a[i]) / 4 / 3 for i in range(4)])
        assert np.allclose(got, want, rtol=1e-6)
        sq = SquishReward(RewardEnv(a))
        assert sq.reward_size == 1 and np.isclose(sq.step(None)[1], a.mean())
    finally:
        update_flags(local_weight=1)


class TickEnv(gym.Env):
    """A stand-in for the bare TrafficEnv surface the Repeater / Remi wrappers touch."""

    class G(object):
        train_roads, intersections = 4, 2

    def __init__(self, done_at=None):
        self.graph = self.G()
        r, i = 4, 2
        self.obs = np.zeros(2 * r + 2 * i, np.int32)
        self.current_phase = self.obs[2 * r:2 * r + i]
        self.elapsed = self.obs[-i:]
        self.rewards = np.zeros(i, np.float32)
        self.reward_size = i
        self.action_space = GSpace([i], np.int32(2))
        self.observation_space = GSpace([2 * r + 2 * i], np.int32(1))
        self.passed_dst = np.zeros(i, bool)
        self.t, self.done_at = 0, done_at

    def _reset(self):
        self.t = 0
        self.obs[:] = 0
        return self.obs

    def _step(self, action):
        self.t += 1
        a = np.asarray(action).astype(np.int32)
        flip = self.current_phase != a
        self.current_phase[:] = a
        self.elapsed[:] = (self.elapsed + 1) * (~flip)
        self.obs[:4] = (self.t + np.arange(4)) % 3
        self.obs[4:8] = (2 * self.t + np.arange(4)) % 5
        self.rewards[:] = -10.0 * (self.t % 4 == 0)
        self.passed_dst[:] = True
        return self.obs, self.rewards, self.t == self.done_at, None

    def remi_reward(self):
        self.rewards[:] = [0.5, -0.5]
        return self.rewards


def test_repeater_and_remi_on_a_scripted_env():
    env = Repeater(5)(TickEnv(done_at=13))
    assert env.observation_space.shape == [10] and env.observation_space.limit.dtype == np.float32
    np.random.seed(0)
    first = env.reset()                      # reset() = inner reset + one decision under a sampled action
    assert first.dtype == np.float32 and env.unwrapped.t == 5
    act = 1 - env.unwrapped.current_phase.copy()
    obs, rew, done, info = env.step(act)     # ticks 6..10, lights flipped at tick 6
    ts = np.arange(6, 11)
    assert np.array_equal(obs[:4], sum((t + np.arange(4)) % 3 for t in ts))
    assert np.array_equal(obs[4:8], (2 * 10 + np.arange(4)) % 5)
    assert np.allclose(obs[8:], 4 / 100 * (2 * act - 1)) and info is None
    assert np.array_equal(rew, [-10.0, -10.0]) and not done           # tick 8
    obs, rew, done, _ = env.step(act)        # breaks at tick 13
    assert done and env.unwrapped.t == 13 and np.array_equal(rew, [-10.0, -10.0])   # tick 12
    try:
        update_flags(mode='validate')
        venv = Repeater(3)(TickEnv())
        venv.reset()
        held = venv.unwrapped.current_phase.copy()
        info = venv.step(np.array([1 - held[0], held[1]]))[3]
        # light 0 had kept its phase for the 3 ticks of the reset decision: elapsed 2 -> (2+1)/2 s
        assert np.allclose(info['light_times'], [1.5])
        assert venv.step(venv.unwrapped.current_phase.copy())[3]['light_times'].size == 0
    finally:
        update_flags(mode='train')
    shaped = Remi(Repeater(2)(TickEnv()))
    shaped.reset()
    obs, rew, done, _ = shaped.step(np.zeros(2, np.int32))
    assert np.array_equal(rew, [0.5, -0.5]) and not shaped.unwrapped.passed_dst.any()


# ---- batched wrappers: per env they equal the single-env wrappers --------------------------------
torch = pytest.importorskip("torch")
from gym_traffic.wrappers import vec as V  # noqa: E402


class CounterVec(object):
    """E CounterEnv-like envs on CPU tensors, each shifted by its env index; env e is done at
    tick done_at[e]."""

    def __init__(self, E, done_at=None, L=6, I=3):
        self.num_envs, self.L, self.I, self.r = E, L, I, 2
        self.action_shape = (E, I)
        self.done_at = done_at
        self.t = 0

    def _obs(self, a):
        k = torch.arange(self.L)[None, :]
        e = torch.arange(self.num_envs)[:, None]
        return ((7 * self.t + 3 * k + 5 * e + a.sum(dim=1, keepdim=True)) % 11).to(torch.int32)

    def reset(self):
        self.t = 0
        return self._obs(torch.zeros(self.action_shape, dtype=torch.int32))

    def step(self, actions, n_ticks=1):
        self.t += 1
        j = torch.arange(self.I)[None, :]
        e = torch.arange(self.num_envs)[:, None]
        rew = ((self.t + j + e + actions) / 4).float()
        done = torch.zeros(self.num_envs, dtype=torch.uint8)
        if self.done_at is not None:
            done = (torch.as_tensor(self.done_at) == self.t).to(torch.uint8)
        return self._obs(actions), rew, done


def single_counter(e, done_at=None):
    """The single-env twin of env e of CounterVec."""
    class One(gym.Env):
        def __init__(self):
            self.observation_space = GSpace([6], np.int32(11))
            self.action_space = GSpace([3], np.int32(2))
            self.reward_size, self.t = 3, 0

        def _o(self, a):
            return ((7 * self.t + 3 * np.arange(6) + 5 * e + int(np.sum(a))) % 11).astype(np.int32)

        def _reset(self):
            self.t = 0
            return self._o(np.zeros(3))

        def _step(self, a):
            self.t += 1
            return self._o(a), ((self.t + np.arange(3) + e + np.asarray(a)) / 4).astype(np.float32), \
                self.t == done_at, None
    return One()


def test_vec_history_and_strobe_equal_single_env_wrappers():
    E = 4
    done_at = [None, 8, None, 3]
    acts = torch.tensor(np.random.RandomState(3).randint(2, size=(5, E, 3)), dtype=torch.int32)

    vh = V.VecHistory(CounterVec(E), 3)
    vh.sample_actions = lambda: torch.ones((E, 3), dtype=torch.int32)
    singles = [HistoryWrapper(3)(single_counter(e)) for e in range(E)]
    for s in singles:
        s.env.action_space.sample = lambda: np.ones(3, np.int32)
    got = vh.reset()
    for e, s in enumerate(singles):
        assert np.array_equal(got[e].numpy(), s.reset())
    for k in range(5):
        got = vh.step(acts[k])
        for e, s in enumerate(singles):
            o, r, d, _ = s.step(acts[k, e].numpy())
            assert np.array_equal(got[0][e].numpy(), o) and np.allclose(got[1][e].numpy(), r)

    vs = V.VecStrobe(CounterVec(E, done_at=[-1 if d is None else d for d in done_at]), 6, 3, [0, 2])
    singles = [StrobeWrapper(6, 3, [0, 2])(single_counter(e, done_at[e])) for e in range(E)]
    for s in singles:
        s.env.reset()
    vs.venv.reset()
    rows, total, done, valid = vs.step(acts[0])
    for e, s in enumerate(singles):
        o, r, d, _ = s.step(acts[0, e].numpy())
        assert int(valid[e]) == len(o) and bool(done[e]) == bool(d)
        assert np.array_equal(rows[e, :len(o)].numpy(), o) and np.allclose(total[e].numpy(), r)


def test_vec_repeater_loop_and_reward_wrappers():
    class TickVec(object):
        def __init__(self, E):
            self.num_envs, self.r, self.I = E, 4, 2
            self.action_shape = (E, 2)
            self.envs = [TickEnv(done_at=7 if e == 1 else None) for e in range(E)]

        def reset(self):
            return torch.as_tensor(np.stack([s.reset().copy() for s in self.envs]))

        def step(self, actions, n_ticks=1):
            out = [s.step(actions[e].numpy()) for e, s in enumerate(self.envs)]
            return (torch.as_tensor(np.stack([o[0].copy() for o in out])),
                    torch.as_tensor(np.stack([o[1].copy() for o in out])),
                    torch.as_tensor(np.array([o[2] for o in out], np.uint8)))

        def remi_reward(self):
            return torch.as_tensor(np.stack([s.remi_reward().copy() for s in self.envs]))

    E = 3
    vec = V.VecRemiRepeater(TickVec(E), 5, remi=False)
    singles = [Repeater(5)(TickEnv(done_at=7 if e == 1 else None)) for e in range(E)]
    vec.venv.reset()
    for s in singles:
        s.env.reset()
    for k in range(2):
        act = torch.tensor([[k, 1 - k]] * E, dtype=torch.int32)
        obs, rew, done = vec.step(act)
        for e, s in enumerate(singles):
            o, r, d, _ = s.step(act[e].numpy())
            assert np.array_equal(obs[e].numpy(), o), (k, e)
            assert np.allclose(rew[e].numpy(), r) and bool(done[e]) == bool(d)
    a = torch.tensor([[1.0, -2.0, 4.0, 0.5]])

    class R(object):
        num_envs = 1

        def step(self, actions):
            return torch.zeros(1, 2), a, torch.zeros(1)
    got = V.VecLocalize(R(), 3).step(None)[1][0].numpy()
    want = [(float(a.sum()) + 2 * float(a[0, i])) / 4 / 3 for i in range(4)]
    assert np.allclose(got, want, rtol=1e-6)
    assert np.isclose(float(V.VecSquish(R()).step(None)[1][0]), float(a.mean()))
