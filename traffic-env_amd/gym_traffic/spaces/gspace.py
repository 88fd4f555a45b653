"""GSpace: the shape-plus-limit space the agents use for multi-intersection actions/observations.

Same surface as the reference's gym_traffic/spaces/gspace.py:4-22 (`shape` list, `size`, `limit` a
NumPy scalar whose dtype is the element type; sample/contains/empty/to_action/replicated), read by
a3c.py:12,22,112-114, qlearn.py:13-31, cem.py:48, greedy.py:16.
"""
import gym
import numpy as np


class GSpace(gym.Space):
    def __init__(self, shape, l):
        self.shape = shape
        self.limit = l
        n = 1
        for dim in shape:
            n *= int(dim)
        self.size = n

    @property
    def dtype(self):
        return self.limit.dtype

    def sample(self):
        # global NumPy RNG on purpose: TrafficEnv._reset draws the initial phases from it
        return np.random.randint(self.limit, size=self.shape, dtype=self.dtype)

    def contains(self, x):
        return x.shape == self.shape

    def empty(self):
        return np.empty(self.shape, dtype=self.dtype)

    def to_action(self, a):
        return np.asarray(a).reshape(self.shape).astype(self.dtype)

    def replicated(self, n):
        return GSpace([n] + list(self.shape), self.limit)

    def __repr__(self):
        return "GSpace(%r, limit=%r)" % (self.shape, self.limit)
