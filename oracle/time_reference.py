#!/usr/bin/env python3
"""Time the reference itself (build container only): vehicle-updates/s of the reference TrafficEnv
imported in place with the identity-`jit` shim (pure NumPy, no Numba, one core) - the "shim-mode
lower bound" of SURVEY.md 8d.  TEST INFRASTRUCTURE: prints numbers for DESIGN.md, nothing imports it.

Usage:  python oracle/time_reference.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402


def run(mods, m, n, L, C, lcps, warm, ticks):
    te = mods["gym_traffic.envs.traffic_env"]
    rg = mods["gym_traffic.envs.roadgraph"]
    args = mods["args"]
    te.CAPACITY = C
    args.update_flags(poisson=True, rate=0.5, local_cars_per_sec=lcps, entry='all', learn_switch=False, mode='train')
    env = mods["gym"].make('traffic-v0')
    env.set_graph(rg.GridRoad(m, n, L))
    env.seed_generator(0)
    env.reset_entrypoints()
    np.random.seed(0)
    env.state[:] = 0
    env.reset()
    I = env.graph.intersections
    act = np.zeros(I, np.int32)
    for t in range(warm):
        if t % 20 == 0:
            act = 1 - act
        env.step(act)
    updates = 0
    t0 = time.perf_counter()
    for t in range(ticks):
        if t % 20 == 0:
            act = 1 - act
        updates += int(np.sum(te.cars_on_roads(env.leading, env.lastcar)))
        env.step(act)
    dt = time.perf_counter() - t0
    return updates / dt, ticks / dt, updates / ticks


def main():
    mods = load_reference()
    for name, a in (("cfg0  2x2  C=10 ", (2, 2, 250.0, 10, 0.12, 100, 2000)),
                    ("3x3 default C=20", (3, 3, 250.0, 20, 0.12, 100, 1500)),
                    ("cfg2 one env 16x16 C=66", (16, 16, 400.0, 66, 0.25, 300, 300))):
        vu, ts, cars = run(mods, *a)
        print("%-26s %10.3e vehicle-updates/s  %8.1f env-steps/s  (%.0f live cars/tick)" % (name, vu, ts, cars))


if __name__ == "__main__":
    main()
