"""CounterEnv: a tiny deterministic gym env used to pin the WRAPPERS (History / Strobe / Last /
Warmup / GSpace adapters) against the reference's own wrapper classes.

TEST INFRASTRUCTURE (oracle/gen_golden_wrappers.py drives the reference's wrappers over it; tests/
drive this repository's wrappers over the very same env and compare with the stored outputs).  It
subclasses whatever `gym.Env` is installed at import time and takes the GSpace class to use, so the
same file serves both sides.

Step t (1-based) under action a:  obs[k] = (7 t + 3 k + sum(a)) mod 11,
reward[j] = (t + j + a[j]) / 4,  done = (t == done_at).  `alias=True` returns one live observation
buffer from every call, like TrafficEnv does (traffic_env.py:248).
"""
import numpy as np


def make_counter_env(gym, GSpace, obs_len=6, n_agents=3, done_at=None, alias=False, array_limit=False,
                     dtype=np.int32):
    class CounterEnv(gym.Env):
        def __init__(self):
            limit = np.full(obs_len, 11, dtype) if array_limit else dtype(11)
            self.observation_space = GSpace([obs_len], limit)
            self.action_space = GSpace([n_agents], np.int32(2))
            self.reward_size = n_agents
            self.t = 0
            self.buf = np.zeros(obs_len, dtype)
            self.resets = 0

        def _obs(self, a):
            vals = (7 * self.t + 3 * np.arange(obs_len) + int(np.sum(a))) % 11
            if alias:
                self.buf[:] = vals
                return self.buf
            return vals.astype(dtype)

        def _reset(self):
            self.t = 0
            self.resets += 1
            return self._obs(np.zeros(n_agents, np.int32))

        def _step(self, action):
            self.t += 1
            a = np.asarray(action).reshape(-1)[:n_agents].astype(np.int64)
            reward = ((self.t + np.arange(n_agents) + a) / 4).astype(np.float32)
            return self._obs(a), reward, self.t == done_at, None

    return CounterEnv()
