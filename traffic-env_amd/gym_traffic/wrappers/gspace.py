"""Adapters between GSpace envs and classic Box/Discrete envs (reference
gym_traffic/wrappers/gspace.py:8-34).

GSpaceWrapper lets the agents (which speak GSpace) drive a classic env such as CartPole:
observations are reshaped to the Box's shape with limit min(high), the action space is one choice
among `n`, the reward becomes a length-1 array.  UnGSpaceWrapper is the opposite direction, used by
`--single_agent` (traffic_test.py:92): a flat Discrete index is unravelled into the GSpace action
shape and the reward vector is averaged.  Like the reference, UnGSpaceWrapper's index space is
`action_space.size` wide and its Box is bounded by the ACTION limit.
"""
import gym
import numpy as np
from gym.spaces import Box, Discrete

from gym_traffic.spaces.gspace import GSpace


class GSpaceWrapper(gym.Wrapper):
    def __init__(self, env):
        super(GSpaceWrapper, self).__init__(env)
        box = env.observation_space
        self.observation_space = GSpace(box.shape, np.float32(np.min(box.high)))
        self.action_space = GSpace([1], np.int32(env.action_space.n))

    def _shaped(self, obs):
        return np.reshape(np.array(obs), self.observation_space.shape)

    def _reset(self):
        return self._shaped(self.env.reset())

    def _step(self, action):
        obs, reward, done, info = self.env.step(np.asarray(action).item())
        return self._shaped(obs), np.array([reward]), done, info


class UnGSpaceWrapper(gym.Wrapper):
    def __init__(self, env):
        super(UnGSpaceWrapper, self).__init__(env)
        self.action_gspace = env.action_space
        self.observation_gspace = env.observation_space
        self.action_space = Discrete(self.action_gspace.size)
        self.observation_space = Box(0, self.action_gspace.limit, shape=self.observation_gspace.shape)

    def _step(self, action):
        cell = np.unravel_index(action, self.action_gspace.shape)
        obs, reward, done, info = self.env.step(cell)
        return obs, np.mean(reward), done, info
