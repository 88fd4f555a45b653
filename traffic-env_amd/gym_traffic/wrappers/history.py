"""HistoryWrapper: the observation becomes the last `history_count` observations, oldest first.

Reference: gym_traffic/wrappers/history.py:5-26.  Behaviour kept: the frame window is a deque that
`_step` shifts by one (so stepping before the first reset raises IndexError, as there); `_reset`
fills it with the reset observation followed by `history_count - 1` steps under sampled actions; the
observation space is the inner one `replicated(history_count)`.  Frames are stored as handed over by
the inner env - an env that returns a live buffer (the bare TrafficEnv does, traffic_env.py:248)
shows the same aliasing as in the reference; behind `Repeater` every frame is a fresh array.
"""
from collections import deque

import gym
import numpy as np

from gym_traffic.wrappers import preset


class History(gym.Wrapper):
    def __init__(self, env, history_count):
        super(History, self).__init__(env)
        self.history_count = int(history_count)
        self.history = deque()
        self.observation_space = env.observation_space.replicated(self.history_count)

    def _reset(self):
        frames = self.history
        frames.clear()
        frames.append(self.env.reset())
        while len(frames) < self.history_count:
            frames.append(self.env.step(self.env.action_space.sample())[0])
        return np.stack(frames)

    def _step(self, action):
        result = self.env.step(action)
        self.history.popleft()
        self.history.append(result[0])
        return (np.array(self.history),) + tuple(result[1:])


def HistoryWrapper(history_count):
    return preset(History, 'HistoryWrapper', history_count=history_count)
