"""Minimal 2017-era `gym` surface (Env/Wrapper/RewardWrapper/Space/make) used ONLY
by oracle/ref_loader.py so that the reference package imports in the build
container (gym is not installed).  TEST INFRASTRUCTURE."""
import importlib

from . import spaces  # noqa: F401
from .envs import registration  # noqa: F401


class Space(object):
    pass


class Env(object):
    metadata = {}
    action_space = None
    observation_space = None

    def step(self, action):
        return self._step(action)

    def reset(self):
        return self._reset()

    def render(self, mode='human', close=False):
        return self._render(mode=mode, close=close)

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space

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
        o, r, d, i = self.env.step(action)
        return o, self._reward(r), d, i


def make(env_id):
    mod, cls = registration.registry[env_id].split(':')
    return getattr(importlib.import_module(mod), cls)()
