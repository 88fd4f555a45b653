"""The wrapper stack every agent of the reference trains against (reference traffic_test.py:27-95):
`Repeater` (one decision = `light_iterations` ticks), `Remi` (reward shaping), `LocalizeWrapper`,
`SquishReward`, and `make_env()` which assembles them from the flags.

`Repeater` over the bare TrafficEnv runs the whole decision as one device submission
(`TrafficEnv.repeat` -> tfx_agent_step, a replayed HIP graph) unless the env is rendering or
`fused=False`; over anything else it is the plain loop.  Both produce the same numbers (tested).
"""
import gym
import numpy as np

from gym_traffic.flags import FLAGS, flag
from gym_traffic.spaces.gspace import GSpace
from gym_traffic.wrappers import preset


class RepeaterBase(gym.Wrapper):
    """Hold one action for `repeat_count` ticks.  Observation float32 [2r + I]:
    [cars that passed each train road, summed | cars near each road's end at the last tick |
    elapsed/100 per intersection, negative while its phase is 0]; reward = sum over the ticks; stops
    at the tick that reports done.  In validate mode `info['light_times']` lists, in seconds, how
    long each light that this action flips had been in its phase (traffic_test.py:41-46)."""

    def __init__(self, env, repeat_count, fused=True):
        super(RepeaterBase, self).__init__(env)
        self.repeat_count = int(repeat_count)
        self.fused = bool(fused)
        graph = self.unwrapped.graph
        self.r, self.i = graph.train_roads, graph.intersections
        self.observation_space = GSpace([2 * self.r + self.i], np.float32(1))

    def _reset(self):
        self.env.reset()
        return self._step(self.action_space.sample())[0]

    def _light_times(self, action):
        flips = np.logical_xor(self.env.current_phase, action).astype(np.int32)
        secs = ((self.env.elapsed + 1) * flips).astype(np.float32) / 2
        return secs[np.nonzero(secs)]

    def _step(self, action):
        info = {'light_times': self._light_times(action)} if flag('mode', 'train') == 'validate' else None
        inner = self.env
        if self.fused and hasattr(inner, 'repeat') and inner is self.unwrapped and not inner.rendering:
            obs, reward, done = inner.repeat(action, self.repeat_count)
            return obs, reward, done, info
        r, i = self.r, self.i
        total_obs = np.zeros(self.observation_space.shape, dtype=np.float32)
        total_reward, done = 0, False
        for _ in range(self.repeat_count):
            obs, reward, done, _ = inner.step(action)
            total_obs[:r] += obs[:r]
            total_obs[r:2 * r] = obs[r:2 * r]
            sign = 2 * obs[-2 * i:-i] - 1
            total_obs[-i:] = obs[-i:] / 100 * sign
            total_reward += reward
            if done:
                break
        return total_obs, total_reward, done, info


def Repeater(repeat_count, fused=True):
    return preset(RepeaterBase, 'Repeater', repeat_count=repeat_count, fused=fused)


class Remi(gym.Wrapper):
    """Replace the reward of a decision by the env's `remi_reward()` (+-0.5 per incoming road from
    waiting / passed flags, traffic_env.py:64-78) and clear `passed_dst` (traffic_test.py:59-64)."""

    def _step(self, action):
        obs, _, done, info = self.env.step(action)
        base = self.unwrapped
        reward = base.remi_reward()
        base.passed_dst[:] = False
        return obs, reward, done, info


class LocalizeWrapper(gym.RewardWrapper):
    """Each intersection's reward becomes the mean of all rewards with its own weighted
    `local_weight` times (traffic_test.py:66-69)."""

    def _reward(self, a):
        w = flag('local_weight', 1)
        return np.mean(np.diag(a) * (w - 1) + a, axis=1) / w


class SquishReward(gym.RewardWrapper):
    """One scalar reward: the mean over intersections (traffic_test.py:71-76)."""

    def __init__(self, env):
        super(SquishReward, self).__init__(env)
        self.reward_size = 1

    def _reward(self, a):
        return np.mean(a)


def make_env(m=3, n=3, length=250, seed=None, capacity=None):
    """The env the reference's drivers build (traffic_test.py:78-93), from the same flags:
    light_secs / rate -> ticks per decision, warmup_lights, remi, local_weight, squish_rewards,
    history, single_agent, render."""
    from gym_traffic.envs.roadgraph import GridRoad
    from gym_traffic.wrappers.history import HistoryWrapper
    from gym_traffic.wrappers.warmup import WarmupWrapper
    from gym_traffic.wrappers.gspace import UnGSpaceWrapper
    env = gym.make('traffic-v0')
    if capacity is not None:
        env.capacity = capacity
    env.set_graph(GridRoad(m, n, length))
    env.seed_generator(seed)
    env.reset_entrypoints()
    if flag('render', False):
        env.rendering = True
    ticks = flag('light_iterations', None)
    if ticks is None:
        ticks = int(flag('light_secs', 5) / FLAGS.rate)
    env = Repeater(ticks)(env)
    if flag('warmup_lights', 0) > 0:
        env = WarmupWrapper(flag('warmup_lights', 0))(env)
    if flag('remi', True):
        env = Remi(env)
    if flag('local_weight', 1) > 1:
        env = LocalizeWrapper(env)
    if flag('squish_rewards', False):
        env = SquishReward(env)
    if flag('history', 1) > 1:
        env = HistoryWrapper(flag('history', 1))(env)
    if flag('single_agent', False):
        env = UnGSpaceWrapper(env)
    return env
