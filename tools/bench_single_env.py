"""Latency of the single-env drop-in surface: gym.make('traffic-v0') stepped tick by tick, and one
agent decision (Repeater(10) + Remi) through the looped and the fused path (wrappers/agent.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import gym_traffic  # noqa: E402,F401
import gym  # noqa: E402
from gym_traffic.envs.roadgraph import GridRoad  # noqa: E402
from gym_traffic.wrappers.agent import Repeater, Remi  # noqa: E402


def make(m, n, length, fused=None):
    env = gym.make('traffic-v0')
    env.set_graph(GridRoad(m, n, length), capacity=66)      # roomy rings: no early `done` in the timing loop
    env.seed_generator(0)
    env.reset_entrypoints()
    if fused is None:
        return env
    return Remi(Repeater(10, fused=fused)(env))


def timeit(fn, n):
    fn(0)
    t0 = time.perf_counter()
    for k in range(n):
        fn(k)
    return (time.perf_counter() - t0) / n * 1e6


for m, n, L in ((3, 3, 250.0), (16, 16, 400.0)):
    np.random.seed(0)
    env = make(m, n, L)
    env.reset()
    a = env.action_space.sample()
    us_tick = timeit(lambda k: env.step(a if (k // 20) % 2 else 1 - a), 300)
    out = ["%dx%d: env.step %.0f us/tick" % (m, n, us_tick)]
    for fused in (False, True):
        w = make(m, n, L, fused)
        w.reset()
        a = w.action_space.sample()
        out.append("%s decision(10 ticks) %.0f us" % ("fused" if fused else "looped", timeit(lambda k: w.step(a if (k // 2) % 2 else 1 - a), 60)))
    print("; ".join(out))
