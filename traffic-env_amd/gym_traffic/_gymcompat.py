"""The slice of the 2017 OpenAI-gym API that traffic-env is written against.

The reference subclasses `gym.Env` through the old `_step/_reset/_render` hooks and monkey-patches
`gym.Env.step` (reference gym_traffic/__init__.py:6-18).  That protocol no longer exists in
current gym/gymnasium and neither package is installed on the MI355X image, so this module
provides the handful of names the env, the wrappers and the agents use.  `gym_traffic/__init__.py`
installs it as `sys.modules['gym']` only when no real `gym` can be imported.
"""
import importlib
import sys
import types


class Space(object):
    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError


class Env(object):
    metadata = {}
    action_space = None
    observation_space = None
    # the two attributes the reference adds to gym.Env (gym_traffic/__init__.py:9,13)
    rendering = False
    reward_size = 1

    def step(self, action):
        # frame-skip wrappers must be able to render inner ticks (gym_traffic/__init__.py:6-8)
        if self.rendering:
            self.render()
        return self._step(action)

    def reset(self):
        return self._reset()

    def render(self, mode='human', close=False):
        return self._render(mode=mode, close=close)

    def close(self):
        return None

    def seed(self, seed=None):
        return []

    @property
    def unwrapped(self):
        return self

    def _step(self, action):
        raise NotImplementedError

    def _reset(self):
        raise NotImplementedError

    def _render(self, mode='human', close=False):
        return None


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.metadata = getattr(env, 'metadata', {})
        self.reward_size = env.reward_size  # gym_traffic/__init__.py:14-18

    def _step(self, action):
        return self.env.step(action)

    def _reset(self):
        return self.env.reset()

    def _render(self, mode='human', close=False):
        return self.env.render(mode, close)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class RewardWrapper(Wrapper):
    def _step(self, action):
        obs, reward, done, info = self.env.step(action)
        return obs, self._reward(reward), done, info

    def _reward(self, reward):
        raise NotImplementedError


class ObservationWrapper(Wrapper):
    def _reset(self):
        return self._observation(self.env.reset())

    def _step(self, action):
        obs, reward, done, info = self.env.step(action)
        return self._observation(obs), reward, done, info


class Discrete(Space):
    def __init__(self, n):
        self.n = n


class Box(Space):
    def __init__(self, low, high, shape=None):
        self.low, self.high, self.shape = low, high, shape


_registry = {}


def register(id, entry_point=None, **kwargs):
    _registry[id] = entry_point


def make(id):
    entry = _registry[id]
    if callable(entry):
        return entry()
    mod, _, cls = entry.partition(':')
    return getattr(importlib.import_module(mod), cls)()


def install():
    """Expose this module as `gym` (+ gym.spaces, gym.envs.registration)."""
    me = sys.modules[__name__]
    gym = types.ModuleType('gym')
    for name in ('Space', 'Env', 'Wrapper', 'RewardWrapper', 'ObservationWrapper', 'make', 'register'):
        setattr(gym, name, getattr(me, name))
    gym.__tfx_compat__ = True
    spaces = types.ModuleType('gym.spaces')
    spaces.Discrete, spaces.Box, spaces.Space = Discrete, Box, Space
    envs = types.ModuleType('gym.envs')
    registration = types.ModuleType('gym.envs.registration')
    registration.register = register
    registration.registry = _registry
    envs.registration = registration
    gym.spaces, gym.envs = spaces, envs
    sys.modules['gym'] = gym
    sys.modules['gym.spaces'] = spaces
    sys.modules['gym.envs'] = envs
    sys.modules['gym.envs.registration'] = registration
    return gym
