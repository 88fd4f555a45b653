"""StrobeWrapper / LastWrapper: frame-skip wrappers that hold one action for `repeat_count` inner
steps (reference gym_traffic/wrappers/strobe.py:5-56).

StrobeWrapper splits the repeat into `num_samples` windows of `repeat_count // num_samples` inner
steps and returns one row per window.  Within a window, the components listed in `sum_indices` are
summed over the window's steps and all others hold the window's LAST observation (each later step
multiplies the row by the 0/1 mask, then adds the new observation).  Rewards are summed.  If the
episode ends inside the repeat, only the rows of the windows completed so far are returned.
`_reset` resets the inner env and returns the observation of one strobe step under a sampled action.

Divergence: the reference builds its mask with `np.zeros_like(observation_space.limit)`, which for
the scalar `limit` every GSpace in the code base carries is 0-dimensional, so its constructor raises
IndexError on `mask[sum_indices] = 1`.  Here a scalar limit yields a mask of the observation's shape
(the evident intent); an array-valued limit is handled exactly as in the reference.

LastWrapper returns the last inner observation and the summed reward; like the reference it does NOT
stop early when the episode ends inside the repeat.
"""
import gym
import numpy as np

from gym_traffic.wrappers import preset


class Strobe(gym.Wrapper):
    def __init__(self, env, repeat_count, num_samples, sum_indices=()):
        super(Strobe, self).__init__(env)
        self.repeat_count = int(repeat_count)
        self.sample_size = self.repeat_count // int(num_samples)
        assert self.sample_size * num_samples == self.repeat_count
        self.observation_space = env.observation_space.replicated(num_samples)
        self.history = self.observation_space.empty()
        limit = np.asarray(env.observation_space.limit)
        self.mask = np.zeros(limit.shape if limit.ndim else tuple(env.observation_space.shape), limit.dtype)
        self.mask[list(sum_indices)] = 1

    def _step(self, action):
        rows, width = self.history, self.sample_size
        total_reward, done, info = 0, False, None
        for k in range(self.repeat_count):
            obs, reward, done, info = self.env.step(action)
            total_reward += reward
            row = k // width
            if k % width == 0:
                rows[row] = obs
            else:
                rows[row] *= self.mask
                rows[row] += obs
            if done:
                return rows[:(k + 1) // width], total_reward, done, info
        return rows, total_reward, done, info

    def _reset(self):
        self.env.reset()
        return self.step(self.env.action_space.sample())[0]


class Last(gym.Wrapper):
    def __init__(self, env, repeat_count):
        super(Last, self).__init__(env)
        self.repeat_count = int(repeat_count)

    def _step(self, action):
        total_reward = 0
        for _ in range(self.repeat_count):
            obs, reward, done, info = self.env.step(action)
            total_reward += reward
        return obs, total_reward, done, info


def StrobeWrapper(repeat_count, num_samples, sum_indices=()):
    return preset(Strobe, 'StrobeWrapper', repeat_count=repeat_count, num_samples=num_samples,
                  sum_indices=sum_indices)


def LastWrapper(repeat_count):
    return preset(Last, 'LastWrapper', repeat_count=repeat_count)
