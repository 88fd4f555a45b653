"""cfg4 closed loop (64x64, 128-car roads, on-device Poisson + greedy) for E envs: env-ticks/s of `calls` tfx_step(50)
calls from an empty start.  usage: c4_loop.py [E] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import numpy as np, torch
from gym_traffic.core import TfxEngine
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
m = n = int(os.environ.get("C4_M", "64"))
eng = TfxEngine(m, n, 800.0, 130, n_envs=E, planes=2)
eng.reset(np.zeros((1, eng.I), np.int32))
eng.set_poisson(0.12 * m * 4 * 0.5, seed=1234)
eng.set_greedy(3)
eng.step(50); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls): eng.step(50)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("cfg4 closed loop E=%d: %.0f env-ticks/s (%.1f us per tick), kernel %s, %d cars on the roads"
      % (E, E * 50 * calls / dt, dt / (50 * calls) * 1e6, eng.step_kernel(), int(eng.cars_on_roads_flat().sum())), flush=True)
