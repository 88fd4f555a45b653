import os, sys, time
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import numpy as np, torch
from gym_traffic.core import TfxEngine
m = n = 64
eng = TfxEngine(m, n, 800.0, 130, n_envs=1, planes=2)
eng.reset(np.zeros((1, eng.I), np.int32))
eng.set_poisson(0.12 * m * 4 * 0.5, seed=1234)
eng.set_greedy(3)
eng.step(50); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): eng.step(50)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("E=1: %.0f env-ticks/s (%.1f us per tick) cars %d" % (2000 / dt, dt / 2000 * 1e6, int(eng.cars_on_roads_flat().sum())))
