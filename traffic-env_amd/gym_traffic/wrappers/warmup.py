"""WarmupWrapper: `reset()` additionally runs `ignore_count` steps under sampled actions and
returns the last observation, so that episodes start from a populated road network.

Reference: gym_traffic/wrappers/warmup.py:3-14 (an episode that ends during warm-up is an
AssertionError there and here).
"""
import gym

from gym_traffic.wrappers import preset


class Warmup(gym.Wrapper):
    def __init__(self, env, ignore_count):
        super(Warmup, self).__init__(env)
        self.ignore_count = int(ignore_count)

    def _reset(self):
        obs = self.env.reset()
        for _ in range(self.ignore_count):
            obs, _, done, _ = self.env.step(self.env.action_space.sample())
            assert not done, "Episode completed during warmup"
        return obs


def WarmupWrapper(ignore_count):
    return preset(Warmup, 'WarmupWrapper', ignore_count=ignore_count)
